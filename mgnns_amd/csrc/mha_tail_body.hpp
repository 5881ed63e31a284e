// The bf16 / split-bf16 fused layer tail (fc + residual + LN + FFN + residual + LN + next layer's w_qs, submodules.py:88-94 and
// 132-139) as a DEVICE FUNCTION over one 16-sample tile, shared by the stand-alone kernel (mha_tail.hip) and the fused
// layer kernel (sq_mha_bf16.hip: the last attention-core workgroup of a tile to finish runs the tile's tail).
#pragma once
#include "common.hpp"
#include "tile_bf16.hpp"

namespace mg_tail {

#ifdef MG_TAIL_TRACE
// profiling aid (off by default): s_memtime stamps of wave 0 of workgroup (0, 0) at the phase boundaries of the tail
__device__ unsigned long long g_tail_trace[16];
#define MG_TSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_tail_trace[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MG_TSTAMP(i) do { } while (0)
#endif

constexpr int NTHR = 512;
constexpr int ROWS = 16;
constexpr int D = 300;                       // d_model (host checks)
constexpr int DT = 19;                       // column tiles of a 300-wide output
constexpr int SD = 322;                      // LDS row stride for 300-wide activations (322 % 32 == 2: conflict-free A frags)
constexpr int SCD = 42;                      // LDS stride (16-B chunks) of 300(->320)-wide bf16 activations

// the same with gamma / beta of this lane's five columns already in registers (loaded at kernel start: inside the
// LayerNorm they are a global round trip right behind the reduction)
__device__ __forceinline__ void ln_rows_r(float* __restrict__ buf, const float (&gamma)[5], const float (&beta)[5], float eps,
                                          int wave, int lane) {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        float* row = buf + (2 * wave + rr) * SD;
        float v[5];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < D ? row[c] : 0.f;
            s += v[i];
        }
        const float mean = wave_sum_dpp(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int c = lane + 64 * i;
            const float d = c < D ? v[i] - mean : 0.f;
            q += d * d;
        }
        const float inv = 1.0f / (sqrtf(wave_sum_dpp(q) / (float)(D - 1)) + eps);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int c = lane + 64 * i;
            if (c < D) row[c] = gamma[i] * (v[i] - mean) * inv + beta[i];
        }
    }
}

// LayerNorm of the tile's rows (two per wave) with the results going straight where the next step wants them, out of the
// registers that hold them: the fp32 row (residual / output), its bf16 hi (+ lo for TERMS == 3) image as the next GEMM's A
// operand, and optionally the global output row -- instead of a second pass over LDS with an integer division per element
// (the phase timer: 7.5 k + 8.5 k cycles of a 53 k-cycle tail for the two LayerNorm + conversion passes).
template <int TERMS>
__device__ __forceinline__ void ln_rows_emit(float* __restrict__ buf, const float (&gamma)[5], const float (&beta)[5], float eps,
                                             int wave, int lane, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo,
                                             bool emit_bf16, float* __restrict__ gout, int rows_valid) {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * wave + rr;
        float* row = buf + r * SD;
        float v[5];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < D ? row[c] : 0.f;
            s += v[i];
        }
        const float mean = wave_sum_dpp(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int c = lane + 64 * i;
            const float d = c < D ? v[i] - mean : 0.f;
            q += d * d;
        }
        const float inv = 1.0f / (sqrtf(wave_sum_dpp(q) / (float)(D - 1)) + eps);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int c = lane + 64 * i;
            if (c < D) {
                const float y = gamma[i] * (v[i] - mean) * inv + beta[i];
                row[c] = y;
                if (emit_bf16) {
                    const unsigned short h = f2bf_t(y);
                    hi[r * SCD * 8 + c] = h;
                    if (TERMS == 3) lo[r * SCD * 8 + c] = f2bf_t(y - bf2f_t(h));
                }
                if (gout && r < rows_valid) gout[(size_t)r * D + c] = y;
            }
        }
    }
}

struct TailW {            // packed hi/lo pairs + fp32 vectors of one layer
    const unsigned short *fc_h, *fc_l, *w1_h, *w1_l, *w2_h, *w2_l, *wq_h, *wq_l;
    const float *fc_b, *g1, *be1, *b1, *b2, *g2, *be2, *bq;
};

// tile: index of the 16-sample tile; crank / csize: this workgroup's rank in / the size of the cluster that shares the tile
// (rank 0 stores `out`, the next layer's projection is split over the ranks).  O_COHERENT: `o` was written by OTHER
// workgroups of the same launch with system-scope write-through stores: read it with loads that bypass the non-coherent
// caches (raw buffer loads, aux sc0 | sc1).
// O_BF16: `o` is already a bf16 matrix [B, HK] (the folded attention's weighted bank rows, sq_mha_folded_bf16.hip): it is copied
// into the A image as it is, any HK that fits LDS (TERMS == 1 only: there is no lo image).
template <int TERMS, bool O_COHERENT, bool O_BF16 = false>
__device__ __forceinline__ void tail_bf16_body(unsigned char* smem_b, const float* __restrict__ o, int HK, const float* __restrict__ q,
                                               int B, const TailW& w, float eps, float* __restrict__ out, int HKn,
                                               float* __restrict__ qh_next, int tile, int crank, int csize,
                                               float* __restrict__ xpart = nullptr, int* __restrict__ xcnt = nullptr) {
    static_assert(!O_BF16 || TERMS == 1, "a bf16 o has no lo part");
    // K-split (with exchange buffers): the ranks of a tile's cluster each contract a slice of the first product's K (1.44 MB of
    // composed weights behind the folded attention, 0.6 MB of fc behind the explicit one) instead of every rank streaming all
    // of it; the last rank to arrive adds the partial sums and finishes the tile (below).
    const bool ksplit = xpart != nullptr && csize > 1;
    const int KSo_all = (HK + 31) / 32;
    const int ks_lo = ksplit ? crank * KSo_all / csize : 0;
    const int ks_hi = ksplit ? (crank + 1) * KSo_all / csize : KSo_all;
    const int so = ksplit ? 4 * ((KSo_all + csize - 1) / csize) + 2 : (HK >> 3) + 2;   // chunk stride of the o tile: == 2 (mod 4)
    uint4* s_oh = reinterpret_cast<uint4*>(smem_b);                 // [16][so]
    uint4* s_ol = s_oh + (O_BF16 ? 0 : ROWS * so);
    uint4* s_ah = s_ol + ROWS * so;                                 // [16][SCD] activation hi (y, h, out in turn)
    uint4* s_al = s_ah + ROWS * SCD;
    float* s_y = reinterpret_cast<float*>(s_al + ROWS * SCD);       // [16][SD] fp32 y (residual of the FFN)
    float* s_t = s_y + ROWS * SD;                                   // [16][SD] fp32 pre-LayerNorm scratch / out
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = tile * ROWS;
    const int KSo = ks_hi - ks_lo;
    constexpr int KSd = (D + 31) / 32;
    MG_TSTAMP(0);

    // per-lane parameter vectors first: b_1 / b_2 of this lane's output columns, gamma / beta of its LayerNorm columns
    const int ccol0 = lane & 15;
    float pb1[3], pb2[3], lg1[5], lb1[5], lg2[5], lb2[5];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = (wave + 8 * t) * 16 + ccol0;
        const bool ok = wave + 8 * t < DT && n < D;
        pb1[t] = ok ? w.b1[n] : 0.f;
        pb2[t] = ok ? w.b2[n] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int c = lane + 64 * i;
        lg1[i] = c < D ? w.g1[c] : 0.f;
        lb1[i] = c < D ? w.be1[c] : 0.f;
        lg2[i] = c < D ? w.g2[c] : 0.f;
        lb2[i] = c < D ? w.be2[c] : 0.f;
    }
    // the first GEMM's weights fly through the staging of o
    WRing<3, TERMS> ring;
    ring_prime(ring, KSo, w.fc_h, w.fc_l, DT, wave, lane, 0, KSo_all, ks_lo);
    // ---- stage o (split), zero the activation buffers (their k-padding must stay zero) ---------------------------
    // every thread's 16-B loads are requested first (the phase timer showed 18 k cycles here with scalar loads consumed
    // item by item: ~30 % of the kernel), converted afterwards
    if (O_BF16) {
        const uint4* ob = reinterpret_cast<const uint4*>(o) + 4 * ks_lo;   // bf16 [B][HK]: HK / 8 chunks per row; this rank's K slice
        const int cpr = HK >> 3, cps = 4 * KSo;
        constexpr int MAXB = (ROWS * (2560 / 8 + 2) + NTHR - 1) / NTHR;     // HK <= 2560; every load requested before the first store
        uint4 v[MAXB];
#pragma unroll
        for (int it = 0; it < MAXB; ++it) {
            const int i = tid + it * NTHR;
            const int r = i / so, c = i - r * so;
            v[it] = make_uint4(0u, 0u, 0u, 0u);
            if (i < ROWS * so && r0 + r < B && c < cps) v[it] = ob[(size_t)(r0 + r) * cpr + c];
        }
#pragma unroll
        for (int it = 0; it < MAXB; ++it) {
            const int i = tid + it * NTHR;
            if (i < ROWS * so) s_oh[i] = v[it];
        }
    } else {
        constexpr int MAXIT = (ROWS * (2048 / 8 + 2) + NTHR - 1) / NTHR;      // HK <= 2048
        const int rows_here = B - r0 < ROWS ? B - r0 : ROWS;
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(o + (size_t)r0 * HK), 0, rows_here * HK * (int)sizeof(float), 0x00027000);
        (void)o_rsrc;
        f32x4 v[MAXIT][2];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int i = tid + it * NTHR;
            const int r = i / so, c = i - r * so;
            v[it][0] = v[it][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int cg = c + 4 * ks_lo;                     // this rank's K slice (all of K without a split)
            if (i < ROWS * so && r0 + r < B && c < 4 * KSo && cg * 8 < HK) {
                if (O_COHERENT) {
                    const int off = (int)(((size_t)r * HK + cg * 8) * sizeof(float));
                    v[it][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(o_rsrc, off, 0, 17));
                    v[it][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(o_rsrc, off + 16, 0, 17));
                } else {
                    const f32x4* src = reinterpret_cast<const f32x4*>(o + (size_t)(r0 + r) * HK + cg * 8);
                    v[it][0] = src[0];
                    v[it][1] = src[1];
                }
            }
        }
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int i = tid + it * NTHR;
            if (i < ROWS * so) {
                uint4 hq, lq = make_uint4(0u, 0u, 0u, 0u);
                const f32x4 x0 = v[it][0], x1 = v[it][1];
                hq.x = f2bf2_t(x0[0], x0[1]); hq.y = f2bf2_t(x0[2], x0[3]); hq.z = f2bf2_t(x1[0], x1[1]); hq.w = f2bf2_t(x1[2], x1[3]);
                if (TERMS == 3) {
                    auto res = [](float x, unsigned int packed, int half) { return x - bf2f_t((unsigned short)(half ? packed >> 16 : packed & 0xFFFFu)); };
                    lq.x = f2bf2_t(res(x0[0], hq.x, 0), res(x0[1], hq.x, 1)); lq.y = f2bf2_t(res(x0[2], hq.y, 0), res(x0[3], hq.y, 1));
                    lq.z = f2bf2_t(res(x1[0], hq.z, 0), res(x1[1], hq.z, 1)); lq.w = f2bf2_t(res(x1[2], hq.w, 0), res(x1[3], hq.w, 1));
                }
                s_oh[i] = hq;
                if (TERMS == 3) s_ol[i] = lq;
            }
        }
    }
    for (int i = tid; i < 2 * ROWS * SCD; i += NTHR) s_ah[i] = make_uint4(0u, 0u, 0u, 0u);
    mg_lds_barrier();        // (LDS only: __syncthreads() would also wait for the weight fragments requested ahead)
    MG_TSTAMP(1);

    unsigned short* ah16 = reinterpret_cast<unsigned short*>(s_ah);
    unsigned short* al16 = reinterpret_cast<unsigned short*>(s_al);
    const int crow = (lane >> 4) * 4, ccol = lane & 15;
    f32x4 acc[3];
    // ---- 1. y = LN1(fc(o) + q) ----------------------------------------------------------------------------------------
    // residual + bias of this lane's 12 outputs requested BEFORE the GEMM (they used to be a global round trip after it)
    f32x4 qb[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = (wave + 8 * t) * 16 + ccol;
        qb[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (wave + 8 * t < DT && n < D) {
            const float bv = w.fc_b[n];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = r0 + crow + r;
                qb[t][r] = bv + (gr < B ? q[(size_t)gr * D + n] : 0.f);
            }
        }
    }
    ring_gemm(acc, ring, s_oh, s_ol, so, KSo, w.fc_h, w.fc_l, lane);
    MG_TSTAMP(2);
    if (ksplit) {
        // partial [tile][rank][wave][t][lane] x 16 B: a lane writes and reads exactly the accumulator slots it owns (write-through
        // stores acknowledged by vmcnt(0), ONE relaxed agent-scope arrival per rank, loads that bypass the non-coherent caches).
        // Nobody waits: the rank that arrives LAST adds the partials in rank order and runs the rest of the tail for the tile
        // alone, the others are done -- the LayerNorm / FFN chain runs once per tile, and there is no co-residency assumption
        // (the first form had every rank wait for the partials and repeat the chain: same speed, 4x the chain's CU time).
        typedef int tl_i32x4 __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t xp = __builtin_amdgcn_make_buffer_rsrc(xpart + (size_t)tile * csize * (8 * 3 * 64 * 4), 0,
                                                                            csize * 8 * 3 * 64 * 16, 0x00027000);
        const int slot = (wave * 3 * 64 + lane) * 16;
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (wave + 8 * t < DT)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tl_i32x4, acc[t]), xp, slot + t * 64 * 16, crank * (8 * 3 * 64 * 16), 17);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's write-through stores are acknowledged
        __syncthreads();
        int* s_flag = reinterpret_cast<int*>(s_t);
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(&xcnt[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == csize - 1;
            if (last) __hip_atomic_store(&xcnt[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            *s_flag = last;
        }
        __syncthreads();
        if (!*s_flag) return;
        __syncthreads();                                       // (s_t is written again below)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
            if (wave + 8 * t < DT) {
                f32x4 pr[8];
#pragma unroll
                for (int rk = 0; rk < 8; ++rk)
                    pr[rk] = rk < csize ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xp, slot + t * 64 * 16, rk * (8 * 3 * 64 * 16), 17))
                                        : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int rk = 0; rk < 8; ++rk) sum += pr[rk];          // rank order, whichever rank arrived last
            }
            acc[t] = sum;
        }
    }
    const int prank = ksplit ? 0 : crank;                              // who stores `out` / how the projection is shared from here on
    const int psize = ksplit ? 1 : csize;
    ring_prime(ring, KSd, w.w1_h, w.w1_l, DT, wave, lane, 0);          // w_1 flies through LayerNorm 1
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = (wave + 8 * t) * 16 + ccol;
        if (wave + 8 * t < DT && n < D) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s_y[(crow + r) * SD + n] = acc[t][r] + qb[t][r];
        }
    }
    mg_lds_barrier();
    ln_rows_emit<TERMS>(s_y, lg1, lb1, eps, wave, lane, ah16, al16, true, nullptr, 0);
    mg_lds_barrier();
    MG_TSTAMP(3);
    // ---- 2. h = relu(w_1 y + b_1) ----------------------------------------------------------------------------------------
    ring_gemm(acc, ring, s_ah, s_al, SCD, KSd, w.w1_h, w.w1_l, lane);
    MG_TSTAMP(4);
    ring_prime(ring, KSd, w.w2_h, w.w2_l, DT, wave, lane, 0);          // w_2 flies through the ReLU / barrier
    mg_lds_barrier();                                   // all A reads of y done before h overwrites the buffer
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = (wave + 8 * t) * 16 + ccol;
        if (wave + 8 * t < DT && n < D) {
            const float bv = pb1[t];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = fmaxf(acc[t][r] + bv, 0.f);
                if (TERMS == 3) split_store(ah16, al16, (crow + r) * SCD * 8 + n, x);
                else ah16[(crow + r) * SCD * 8 + n] = f2bf_t(x);
            }
        }
    }
    mg_lds_barrier();
    MG_TSTAMP(5);
    // ---- 3. out = LN2(w_2 h + b_2 + y) --------------------------------------------------------------------------------------
    ring_gemm(acc, ring, s_ah, s_al, SCD, KSd, w.w2_h, w.w2_l, lane);
    MG_TSTAMP(6);
    // the projection's first column-tile pair flies through LayerNorm 2
    const int NTq = (HKn + 15) / 16;
    const int slots = (NTq + 7) / 8;                                       // column-tile slots per wave over the whole N
    const int per = (slots + psize - 1) / psize;
    const int s_lo = prank * per, s_hi = min(slots, s_lo + per);
    // TERMS == 1: the projection's weights as ONE stream per wave -- k-step d of the wave's next column tile is requested the
    // moment k-step d of the current one is consumed (a ring of KSd fragments), so the stream never restarts cold (with a ring
    // primed per pair of column tiles every pair paid an L2 round trip: 21 us for the 1.44 MB composed query map split over
    // two ranks).  TERMS == 3 keeps the pairwise ring (hi + lo fragments of a stream do not fit the register budget).
    WRing<2, TERMS> ringq;
    uint4 rq[KSd];
    const uint4* Wq = reinterpret_cast<const uint4*>(w.wq_h);
    auto wq_off = [&](int slot) {
        const int nt = wave + 8 * slot;
        return ((size_t)(nt < NTq ? nt : 0) * KSd) * 64 + lane;
    };
    if (TERMS == 1) {
        if (w.wq_h && s_lo < s_hi) {
            const size_t o0 = wq_off(s_lo);
#pragma unroll
            for (int d = 0; d < KSd; ++d) rq[d] = Wq[o0 + (size_t)d * 64];
        }
    } else if (w.wq_h) {
        ring_prime(ringq, KSd, w.wq_h, w.wq_l, NTq, wave, lane, s_lo);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = (wave + 8 * t) * 16 + ccol;
        if (wave + 8 * t < DT && n < D) {
            const float bv = pb2[t];
#pragma unroll
            for (int r = 0; r < 4; ++r) s_t[(crow + r) * SD + n] = acc[t][r] + bv + s_y[(crow + r) * SD + n];
        }
    }
    mg_lds_barrier();
    ln_rows_emit<TERMS>(s_t, lg2, lb2, eps, wave, lane, ah16, al16, w.wq_h != nullptr, prank == 0 ? out + (size_t)r0 * D : nullptr,
                        B - r0);
    MG_TSTAMP(7);
    // ---- 4. next layer's query projection -------------------------------------------------------------------------------------
    // gridDim.y workgroups share a 16-sample tile: each recomputes steps 1-3 (identical results; rank 0 stores `out`) and
    // takes 1 / gridDim.y of the projection's column tiles.  The projection is 40 % of the weight bytes a workgroup streams
    // at the per-CU L2 rate, and the only part of the chain whose columns are independent.
    if (TERMS == 1 && w.wq_h) {
        mg_lds_barrier();
        bf16x8 aq[KSd];
#pragma unroll
        for (int d = 0; d < KSd; ++d) aq[d] = __builtin_bit_cast(bf16x8, s_ah[(lane & 15) * SCD + (lane >> 4) + 4 * d]);
        for (int slot = s_lo; slot < s_hi; ++slot) {
            const int nt = wave + 8 * slot, n = nt * 16 + ccol;
            const bool live = nt < NTq && n < HKn;
            const float bv = (live && w.bq) ? w.bq[n] : 0.f;
            const bool more = slot + 1 < s_hi;
            const size_t nx = wq_off(more ? slot + 1 : slot);
            f32x4 a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int d = 0; d < KSd; ++d) {
                const uint4 c = rq[d];
                if (more) rq[d] = Wq[nx + (size_t)d * 64];
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[d], __builtin_bit_cast(bf16x8, c), a2, 0, 0, 0);
            }
            if (live) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gr = r0 + crow + r;
                    if (gr < B) qh_next[(size_t)gr * HKn + n] = a2[r] + bv;
                }
            }
        }
    } else if (w.wq_h) {
        mg_lds_barrier();
        for (int t0 = s_lo; t0 < s_hi; t0 += 2) {
            f32x4 a2[2];
            if (t0 != s_lo) ring_prime(ringq, KSd, w.wq_h, w.wq_l, NTq, wave, lane, t0);
            ring_gemm(a2, ringq, s_ah, s_al, SCD, KSd, w.wq_h, w.wq_l, lane);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int nt = wave + 8 * (t0 + t);
                const int n = nt * 16 + ccol;
                if (t0 + t < s_hi && nt < NTq && n < HKn) {
                    const float bv = w.bq ? w.bq[n] : 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int gr = r0 + crow + r;
                        if (gr < B) qh_next[(size_t)gr * HKn + n] = a2[t][r] + bv;
                    }
                }
            }
        }
    }
    MG_TSTAMP(8);
}


}  // namespace mg_tail
