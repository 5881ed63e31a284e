// Exact-fp32 MFMA building block shared by the fused row-tile kernels (mha_tail.hip, label_tail.hip): a 16-row tile of
// activations in LDS times a weight matrix streamed from L2 in the fragment-major layout of mgnns_pack_weight_f32:
//   Wp[nt][kq][lane][4]: B fragments of k-steps 4kq..4kq+3 for column tile nt,
//   Wp[...][j] = W[nt*16 + (lane&15)][(4*kq + j)*4 + (lane>>4)]     (0 outside [N, K])
// v_mfma_f32_16x16x4_f32 is bit-equal to an fmaf chain, so these kernels keep fp32 parity.
#pragma once
#include "common.hpp"

// C[16 x N] tile-GEMM of one wave: acc[t] for column tiles nt = wave + 8t (t < TPW), A from LDS (stride sa), K padded
// to a multiple of 16 by zero weights (A beyond K must be finite: buffers are zero padded).
template <int TPW>
__device__ __forceinline__ void mg_tile_gemm_f32(f32x4 (&acc)[TPW], const float* __restrict__ As, int sa, int K,
                                          const float* __restrict__ Wp, int NTt, int wave, int lane, int t0) {
    const int KQ = (K + 15) / 16;
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* ap = As + (lane & 15) * sa + (lane >> 4);
    const f32x4* wp[TPW];
    bool on[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int nt = wave + 8 * (t0 + t);
        on[t] = nt < NTt;
        wp[t] = reinterpret_cast<const f32x4*>(Wp) + ((size_t)(on[t] ? nt : 0) * KQ) * 64 + lane;
    }
    // weight fragments run PF k-quads ahead (they come from L2: ~1 us under load, a k-quad of MFMAs is ~0.15 us)
    constexpr int PF = 4;
    f32x4 ring[PF][TPW];
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
        for (int t = 0; t < TPW; ++t) ring[d][t] = d < KQ ? wp[t][(size_t)d * 64] : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kq0 = 0; kq0 < KQ; kq0 += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int kq = kq0 + d;
            if (kq < KQ) {
                f32x4 cb[TPW];
#pragma unroll
                for (int t = 0; t < TPW; ++t) cb[t] = ring[d][t];
                if (kq + PF < KQ) {
#pragma unroll
                    for (int t = 0; t < TPW; ++t) ring[d][t] = wp[t][(size_t)(kq + PF) * 64];
                }
                float a[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = ap[(4 * kq + j) * 4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < TPW; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], cb[t][j], acc[t], 0, 0, 0);
            }
        }
    }
}


// The same product over the k-quads [kq_lo, kq_hi) of a K that is walked in CHUNKS (A holds only the chunk: its column 0 is
// k = 16 kq_lo), accumulating into acc -- for operands too wide to stage in LDS at once.  KQ = k-quads of the whole K (the
// packed weight's row length).
template <int TPW>
__device__ __forceinline__ void mg_tile_gemm_f32_chunk(f32x4 (&acc)[TPW], const float* __restrict__ As, int sa, int kq_lo, int kq_hi,
                                                       int KQ, const float* __restrict__ Wp, int NTt, int wave, int lane, int t0) {
    const float* ap = As + (lane & 15) * sa + (lane >> 4);
    const f32x4* wp[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int nt = wave + 8 * (t0 + t);
        wp[t] = reinterpret_cast<const f32x4*>(Wp) + ((size_t)(nt < NTt ? nt : 0) * KQ) * 64 + lane;
    }
    constexpr int PF = 4;
    f32x4 ring[PF][TPW];
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
        for (int t = 0; t < TPW; ++t) ring[d][t] = kq_lo + d < kq_hi ? wp[t][(size_t)(kq_lo + d) * 64] : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kq0 = kq_lo; kq0 < kq_hi; kq0 += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int kq = kq0 + d;
            if (kq < kq_hi) {
                f32x4 cb[TPW];
#pragma unroll
                for (int t = 0; t < TPW; ++t) cb[t] = ring[d][t];
                if (kq + PF < kq_hi) {
#pragma unroll
                    for (int t = 0; t < TPW; ++t) ring[d][t] = wp[t][(size_t)(kq + PF) * 64];
                }
                float a[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = ap[(4 * (kq - kq_lo) + j) * 4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < TPW; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], cb[t][j], acc[t], 0, 0, 0);
            }
        }
    }
}
