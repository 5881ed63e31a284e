// bf16 / split-bf16 MFMA building blocks shared by the fused row-tile kernels (mha_tail.hip, label_tail.hip): a 16-row tile of
// activations in LDS (16-B chunks of 8 bf16, hi and optionally lo parts) times a weight matrix streamed from L2 in the
// fragment-major layout of mgnns_pack_weight_bf16_split:
//   Wp{hi,lo}[nt][ks][lane][8] = split(W[nt*16 + (lane&15)][ks*32 + (lane>>4)*8 + j])   (0 outside [N,K])
// TERMS = 1: plain bf16 operands.  TERMS = 3: every fp32 operand x is carried as hi = bf16(x), lo = bf16(x - hi) and a
// product is formed as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_16x16x32_bf16 (fp32 accumulate).
#pragma once
#include "common.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// fp32 -> bf16, round to nearest even: ONE v_cvt_pk_bf16_f32 (there is no builtin for it on gfx950) instead of the five-instruction
// integer form -- these kernels are made of small latency-bound phases in which every VALU instruction shows
__device__ __forceinline__ unsigned short f2bf_t(float x) {
    unsigned int r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(r) : "v"(x));
    return (unsigned short)r;
}
// two values -> packed bf16x2 (lo half = a)
__device__ __forceinline__ unsigned int f2bf2_t(float a, float b) {
    unsigned int r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float bf2f_t(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }
__device__ __forceinline__ void split_store(unsigned short* hi, unsigned short* lo, int idx, float x) {
    const unsigned short h = f2bf_t(x);
    hi[idx] = h;
    lo[idx] = f2bf_t(x - bf2f_t(h));
}

// Weight fragments of a GEMM's first PF k-steps, requested AHEAD of the phase that consumes them: every GEMM of the chain
// used to start cold (request, wait an L2 round trip of ~1 us, compute), four times per layer; the weights do not depend on
// the activations, so the next GEMM's first fragments fly through the LayerNorm / conversion / barrier in front of it.
template <int TPW, int TERMS, int PFX = 0>
struct WRing {
    static constexpr int PF = PFX ? PFX : (TERMS == 1 ? 10 : 3);     // k-steps in flight (L2 latency ~1-2 us, a k-step of MFMAs ~50 ns)
    uint4 rh[PF][TPW], rl[TERMS == 3 ? PF : 1][TPW];
    size_t woff[TPW];
};

template <int TPW, int TERMS, int PFX>
__device__ __forceinline__ void ring_prime(WRing<TPW, TERMS, PFX>& r, int KS, const unsigned short* __restrict__ Whi,
                                           const unsigned short* __restrict__ Wlo, int NTt, int wave, int lane, int t0,
                                           int KSW = 0, int ks_off = 0) {
    // KS: k-steps this GEMM pass walks; KSW: k-steps per column tile in the packed weight (0: = KS); ks_off: first k-step of
    // the pass inside the weight (a K walked in several passes because the activations do not fit LDS at once)
    if (KSW == 0) KSW = KS;
    constexpr int PF = WRing<TPW, TERMS, PFX>::PF;
    const uint4* Wh = reinterpret_cast<const uint4*>(Whi);
    const uint4* Wl = reinterpret_cast<const uint4*>(Wlo);
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int nt = wave + 8 * (t0 + t);
        r.woff[t] = ((size_t)(nt < NTt ? nt : 0) * KSW + ks_off) * 64 + lane;
    }
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            r.rh[d][t] = d < KS ? Wh[r.woff[t] + (size_t)d * 64] : make_uint4(0u, 0u, 0u, 0u);
            if (TERMS == 3) r.rl[d][t] = d < KS ? Wl[r.woff[t] + (size_t)d * 64] : make_uint4(0u, 0u, 0u, 0u);
        }
}

// acc = A[16 x K] . W^T for this wave's TPW column tiles; `r` must have been primed for the same (W, t0).
template <int TPW, int TERMS, int PFX>
__device__ __forceinline__ void ring_gemm(f32x4 (&acc)[TPW], WRing<TPW, TERMS, PFX>& r, const uint4* __restrict__ Ahi,
                                          const uint4* __restrict__ Alo, int sa, int KS, const unsigned short* __restrict__ Whi,
                                          const unsigned short* __restrict__ Wlo, int lane, bool accumulate = false) {
    constexpr int PF = WRing<TPW, TERMS, PFX>::PF;
    if (!accumulate) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int aoff = (lane & 15) * sa + (lane >> 4);
    const uint4* Wh = reinterpret_cast<const uint4*>(Whi);
    const uint4* Wl = reinterpret_cast<const uint4*>(Wlo);
    for (int ks0 = 0; ks0 < KS; ks0 += PF) {
        // the chunk's A fragments (activations in LDS) all at once: read one k-step at a time right before its MFMAs,
        // every k-step exposed an LDS round trip (3 MFMAs per wave and k-step hide nothing)
        uint4 ahq[PF], alq[TERMS == 3 ? PF : 1];
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int ks = ks0 + d < KS ? ks0 + d : KS - 1;
            ahq[d] = Ahi[aoff + ks * 4];
            if (TERMS == 3) alq[d] = Alo[aoff + ks * 4];
        }
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int ks = ks0 + d;
            if (ks < KS) {
                uint4 ch[TPW], cl[TPW];
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    ch[t] = r.rh[d][t];
                    if (TERMS == 3) cl[t] = r.rl[d][t];
                }
                if (ks + PF < KS) {
#pragma unroll
                    for (int t = 0; t < TPW; ++t) {
                        r.rh[d][t] = Wh[r.woff[t] + (size_t)(ks + PF) * 64];
                        if (TERMS == 3) r.rl[d][t] = Wl[r.woff[t] + (size_t)(ks + PF) * 64];
                    }
                }
                const bf16x8 ah = __builtin_bit_cast(bf16x8, ahq[d]);
                bf16x8 al = ah;
                if (TERMS == 3) al = __builtin_bit_cast(bf16x8, alq[d]);
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, ch[t]), acc[t], 0, 0, 0);
                    if (TERMS == 3) {
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, cl[t]), acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, __builtin_bit_cast(bf16x8, ch[t]), acc[t], 0, 0, 0);
                    }
                }
            }
        }
    }
}

