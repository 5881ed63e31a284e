// Everything of a MyMultiHeadAttention layer behind the attention core, in ONE launch (moudles.py:224-225 ->
// submodules.py:88-94 and 132-139), plus the NEXT layer's query projection:
//   y   = LN1(fc(o) + q)                        MultiHeadAttention: fc, residual, custom LayerNorm
//   z   = w_2 relu(w_1 y + b_1) + b_2           PositionwiseFeedForward (Conv1d k=1 == Linear)
//   out = LN2(z + y)
//   qh' = w_qs'(out)                            (optional) w_qs of the following layer of the stack
// The reference runs this as 4 GEMMs + 2 LayerNorms (+ the next w_qs) per layer; at M = batch = 256 every one of
// them is a latency-bound launch.  Here a workgroup owns 16 samples (one MFMA row tile) through the whole
// chain: activations stay in LDS, weights stream from L2 in a pre-packed fragment-major layout (one 16-B load
// = the B fragments of four k-steps), 8 waves split the output column tiles.  Exact fp32 (v_mfma_f32_16x16x4_f32).
#include "common.hpp"

namespace {

constexpr int NTHR = 512;
constexpr int ROWS = 16;
constexpr int D = 300;                       // d_model (host checks)
constexpr int DT = 19;                       // column tiles of a 300-wide output
constexpr int SD = 322;                      // LDS row stride for 300-wide activations (322 % 32 == 2: conflict-free A frags)

// Wp[nt][kq][lane][4]: B fragments of k-steps 4kq..4kq+3 for column tile nt:
//   Wp[...][j] = W[nt*16 + (lane&15)][(4*kq + j)*4 + (lane>>4)]     (0 outside [N, K])
__global__ __launch_bounds__(256) void pack_w_f32_kernel(const float* __restrict__ W, int N, int K, float* __restrict__ Wp) {
    const int KQ = (K + 15) / 16, NTt = (N + 15) / 16;
    const size_t total = (size_t)NTt * KQ * 64;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        size_t r = i >> 6;
        const int kq = (int)(r % KQ);
        const int nt = (int)(r / KQ);
        const int n = nt * 16 + (lane & 15);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = (4 * kq + j) * 4 + (lane >> 4);
            if (n < N && k < K) v[j] = W[(size_t)n * K + k];
        }
        reinterpret_cast<f32x4*>(Wp)[i] = v;
    }
}

// C[16 x N] tile-GEMM of one wave: acc[t] for column tiles nt = wave + 8t (t < TPW), A from LDS (stride sa), K padded
// to a multiple of 16 by zero weights (A beyond K must be finite: buffers are zero padded).
template <int TPW>
__device__ __forceinline__ void tile_gemm(f32x4 (&acc)[TPW], const float* __restrict__ As, int sa, int K,
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

// custom LayerNorm of the 16 rows held in LDS (stride SD), in place; wave w owns rows 2w and 2w+1
__device__ __forceinline__ void ln_rows(float* __restrict__ buf, const float* __restrict__ gamma,
                                        const float* __restrict__ beta, float eps, int wave, int lane) {
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
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int c = lane + 64 * i;
            const float d = c < D ? v[i] - mean : 0.f;
            q += d * d;
        }
        const float inv = 1.0f / (sqrtf(wave_sum(q) / (float)(D - 1)) + eps);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int c = lane + 64 * i;
            if (c < D) row[c] = gamma[c] * (v[i] - mean) * inv + beta[c];
        }
    }
}

__global__ __launch_bounds__(NTHR) void mha_tail_kernel(const float* __restrict__ o, int HK, const float* __restrict__ q, int B,
                                                        const float* __restrict__ fc_wp, const float* __restrict__ fc_b,
                                                        const float* __restrict__ g1, const float* __restrict__ be1,
                                                        const float* __restrict__ w1_wp, const float* __restrict__ b1,
                                                        const float* __restrict__ w2_wp, const float* __restrict__ b2,
                                                        const float* __restrict__ g2, const float* __restrict__ be2, float eps,
                                                        float* __restrict__ out, const float* __restrict__ wq_wp,
                                                        const float* __restrict__ bq, int HKn, float* __restrict__ qh_next) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int so = HK + 2 + ((32 - (HK % 32)) % 32);       // stride of the o tile: == 2 (mod 32)
    float* s_o = smem;                                      // [16][so]
    float* s_y = s_o + ROWS * so;                           // [16][SD]  y, later out
    float* s_h = s_y + ROWS * SD;                           // [16][SD]  relu(w1 y), later z + y
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * ROWS;

    // ---- stage o rows (zero beyond B), zero the pads the k-loops may touch --------------------------------------
    for (int i = tid; i < ROWS * so; i += NTHR) {
        const int r = i / so, c = i - r * so;
        s_o[i] = (r0 + r < B && c < HK) ? o[(size_t)(r0 + r) * HK + c] : 0.f;
    }
    for (int i = tid; i < 2 * ROWS * SD; i += NTHR) s_y[i] = 0.f;
    __syncthreads();

    const int crow = (lane >> 4) * 4, ccol = lane & 15;
    f32x4 acc[3];
    // ---- 1. y = LN1(fc(o) + q) --------------------------------------------------------------------------------------
    tile_gemm<3>(acc, s_o, so, HK, fc_wp, DT, wave, lane, 0);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = (wave + 8 * t) * 16 + ccol;
        if (wave + 8 * t < DT && n < D) {
            const float bv = fc_b[n];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = r0 + crow + r;
                s_y[(crow + r) * SD + n] = acc[t][r] + bv + (gr < B ? q[(size_t)gr * D + n] : 0.f);
            }
        }
    }
    __syncthreads();
    ln_rows(s_y, g1, be1, eps, wave, lane);
    __syncthreads();
    // ---- 2. h = relu(w_1 y + b_1) ------------------------------------------------------------------------------------
    tile_gemm<3>(acc, s_y, SD, D, w1_wp, DT, wave, lane, 0);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = (wave + 8 * t) * 16 + ccol;
        if (wave + 8 * t < DT && n < D) {
            const float bv = b1[n];
#pragma unroll
            for (int r = 0; r < 4; ++r) s_h[(crow + r) * SD + n] = fmaxf(acc[t][r] + bv, 0.f);
        }
    }
    __syncthreads();
    // ---- 3. out = LN2(w_2 h + b_2 + y) ----------------------------------------------------------------------------------
    tile_gemm<3>(acc, s_h, SD, D, w2_wp, DT, wave, lane, 0);
    __syncthreads();                                   // every wave is done reading s_h before it is overwritten
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = (wave + 8 * t) * 16 + ccol;
        if (wave + 8 * t < DT && n < D) {
            const float bv = b2[n];
#pragma unroll
            for (int r = 0; r < 4; ++r) s_h[(crow + r) * SD + n] = acc[t][r] + bv + s_y[(crow + r) * SD + n];
        }
    }
    __syncthreads();
    ln_rows(s_h, g2, be2, eps, wave, lane);
    __syncthreads();
    for (int i = tid; i < ROWS * D; i += NTHR) {
        const int r = i / D, c = i - r * D;
        if (r0 + r < B) out[(size_t)(r0 + r) * D + c] = s_h[r * SD + c];
    }
    // ---- 4. next layer's query projection qh' = w_qs'(out) + b ---------------------------------------------------------------
    if (wq_wp) {
        const int NTq = (HKn + 15) / 16;
        for (int t0 = 0; t0 * 8 < NTq; t0 += 4) {
            f32x4 a4[4];
            tile_gemm<4>(a4, s_h, SD, D, wq_wp, NTq, wave, lane, t0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int nt = wave + 8 * (t0 + t);
                const int n = nt * 16 + ccol;
                if (nt < NTq && n < HKn) {
                    const float bv = bq ? bq[n] : 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int gr = r0 + crow + r;
                        if (gr < B) qh_next[(size_t)gr * HKn + n] = a4[t][r] + bv;
                    }
                }
            }
        }
    }
}

}  // namespace

extern "C" size_t mgnns_packed_f32_weight_bytes(int N, int K) {
    return (size_t)((N + 15) / 16) * ((K + 15) / 16) * 64 * 16;
}

extern "C" int mgnns_pack_weight_f32(const float* W, int N, int K, float* Wp, mgnns_stream_t stream) {
    MG_REQUIRE(W && Wp && N > 0 && K > 0, "mgnns_pack_weight_f32: bad arguments");
    const size_t total = (size_t)((N + 15) / 16) * ((K + 15) / 16) * 64;
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_w_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, W, N, K, Wp);
    MG_CHECK_LAUNCH("mgnns_pack_weight_f32");
    return 0;
}

extern "C" int mgnns_mha_tail_fwd(const float* o, int HK, const float* q, int B, int d_model,
                                  const float* fc_wp, const float* fc_b, const float* ln1_gamma, const float* ln1_beta,
                                  const float* w1_wp, const float* b1, const float* w2_wp, const float* b2,
                                  const float* ln2_gamma, const float* ln2_beta, float eps, float* out,
                                  const float* wq_next_wp, const float* bq_next, int HK_next, float* qh_next,
                                  mgnns_stream_t stream) {
    MG_REQUIRE(o && q && fc_wp && fc_b && ln1_gamma && ln1_beta && w1_wp && b1 && w2_wp && b2 && ln2_gamma && ln2_beta && out,
               "mgnns_mha_tail_fwd: null pointer");
    MG_REQUIRE(d_model == D, "mgnns_mha_tail_fwd: d_model=%d unsupported (300 only)", d_model);
    MG_REQUIRE(HK > 0 && HK % 4 == 0 && HK <= 2048, "mgnns_mha_tail_fwd: n_head*d_v=%d unsupported (multiple of 4, <= 2048)", HK);
    MG_REQUIRE(!wq_next_wp || (qh_next && HK_next > 0), "mgnns_mha_tail_fwd: next-layer projection needs qh_next and HK_next");
    if (B <= 0) return 0;
    const int so = HK + 2 + ((32 - (HK % 32)) % 32);
    const size_t lds = ((size_t)ROWS * so + 2 * (size_t)ROWS * SD) * sizeof(float);
    MG_REQUIRE(lds <= 160 * 1024, "mgnns_mha_tail_fwd: needs %zu B of LDS", lds);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mha_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(mha_tail_kernel, dim3((B + ROWS - 1) / ROWS), dim3(NTHR), lds, (hipStream_t)stream, o, HK, q, B, fc_wp,
                       fc_b, ln1_gamma, ln1_beta, w1_wp, b1, w2_wp, b2, ln2_gamma, ln2_beta, eps, out, wq_next_wp, bq_next,
                       HK_next, qh_next);
    MG_CHECK_LAUNCH("mgnns_mha_tail_fwd");
    return 0;
}
