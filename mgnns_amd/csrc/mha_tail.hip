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
#include "tile_f32.hpp"
#include "tile_bf16.hpp"
#include "mha_tail_body.hpp"

namespace {

using mg_tail::NTHR;
using mg_tail::ROWS;
using mg_tail::D;
using mg_tail::DT;
using mg_tail::SD;
using mg_tail::SCD;
using mg_tail::TailW;
using mg_tail::ln_rows_r;

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

    // per-lane parameter vectors and the residual first (see the bf16 kernel: each was a global round trip behind a GEMM)
    const int crow = (lane >> 4) * 4, ccol = lane & 15;
    float pb1[3], pb2[3], lg1[5], lb1[5], lg2[5], lb2[5];
    f32x4 qb[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = (wave + 8 * t) * 16 + ccol;
        const bool ok = wave + 8 * t < DT && n < D;
        pb1[t] = ok ? b1[n] : 0.f;
        pb2[t] = ok ? b2[n] : 0.f;
        qb[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok) {
            const float bv = fc_b[n];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = r0 + crow + r;
                qb[t][r] = bv + (gr < B ? q[(size_t)gr * D + n] : 0.f);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int c = lane + 64 * i;
        lg1[i] = c < D ? g1[c] : 0.f;
        lb1[i] = c < D ? be1[c] : 0.f;
        lg2[i] = c < D ? g2[c] : 0.f;
        lb2[i] = c < D ? be2[c] : 0.f;
    }
    // ---- stage o rows (zero beyond B), zero the pads the k-loops may touch: 16-B loads, all in flight first -------------
    {
        const int hk4 = HK >> 2;                                   // HK % 4 == 0 (checked by the launcher)
        constexpr int MAXIT = (ROWS * (2048 / 4) + NTHR - 1) / NTHR;
        f32x4 v[MAXIT];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int i = tid + it * NTHR;
            const int r = i / hk4, c4 = i - r * hk4;
            v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < ROWS * hk4 && r0 + r < B) v[it] = *reinterpret_cast<const f32x4*>(o + (size_t)(r0 + r) * HK + 4 * c4);
        }
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int i = tid + it * NTHR;
            if (i < ROWS * hk4) {
                const int r = i / hk4, c4 = i - r * hk4;
                float* d = s_o + r * so + 4 * c4;                  // so is even: 8-byte aligned
                *reinterpret_cast<float2*>(d) = float2{v[it][0], v[it][1]};
                *reinterpret_cast<float2*>(d + 2) = float2{v[it][2], v[it][3]};
            }
        }
        for (int i = tid; i < ROWS * (so - HK); i += NTHR) s_o[(i / (so - HK)) * so + HK + i % (so - HK)] = 0.f;
    }
    for (int i = tid; i < 2 * ROWS * SD; i += NTHR) s_y[i] = 0.f;
    __syncthreads();

    f32x4 acc[3];
    // ---- 1. y = LN1(fc(o) + q) --------------------------------------------------------------------------------------
    mg_tile_gemm_f32<3>(acc, s_o, so, HK, fc_wp, DT, wave, lane, 0);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = (wave + 8 * t) * 16 + ccol;
        if (wave + 8 * t < DT && n < D) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s_y[(crow + r) * SD + n] = acc[t][r] + qb[t][r];
        }
    }
    __syncthreads();
    ln_rows_r(s_y, lg1, lb1, eps, wave, lane);
    __syncthreads();
    // ---- 2. h = relu(w_1 y + b_1) ------------------------------------------------------------------------------------
    mg_tile_gemm_f32<3>(acc, s_y, SD, D, w1_wp, DT, wave, lane, 0);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = (wave + 8 * t) * 16 + ccol;
        if (wave + 8 * t < DT && n < D) {
            const float bv = pb1[t];
#pragma unroll
            for (int r = 0; r < 4; ++r) s_h[(crow + r) * SD + n] = fmaxf(acc[t][r] + bv, 0.f);
        }
    }
    __syncthreads();
    // ---- 3. out = LN2(w_2 h + b_2 + y) ----------------------------------------------------------------------------------
    mg_tile_gemm_f32<3>(acc, s_h, SD, D, w2_wp, DT, wave, lane, 0);
    __syncthreads();                                   // every wave is done reading s_h before it is overwritten
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = (wave + 8 * t) * 16 + ccol;
        if (wave + 8 * t < DT && n < D) {
            const float bv = pb2[t];
#pragma unroll
            for (int r = 0; r < 4; ++r) s_h[(crow + r) * SD + n] = acc[t][r] + bv + s_y[(crow + r) * SD + n];
        }
    }
    __syncthreads();
    ln_rows_r(s_h, lg2, lb2, eps, wave, lane);
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
            mg_tile_gemm_f32<4>(a4, s_h, SD, D, wq_wp, NTq, wave, lane, t0);
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
    MG_DYN_LDS(mha_tail_kernel, 160 * 1024);
    hipLaunchKernelGGL(mha_tail_kernel, dim3((B + ROWS - 1) / ROWS), dim3(NTHR), lds, (hipStream_t)stream, o, HK, q, B, fc_wp,
                       fc_b, ln1_gamma, ln1_beta, w1_wp, b1, w2_wp, b2, ln2_gamma, ln2_beta, eps, out, wq_next_wp, bq_next,
                       HK_next, qh_next);
    MG_CHECK_LAUNCH("mgnns_mha_tail_fwd");
    return 0;
}

// =====================================================================================================================
// bf16-MFMA variant of the same fused tail (used with the bf16 precision mode).  TERMS = 1: plain bf16 operands.
// TERMS = 3: split-bf16 -- every fp32 operand x is carried as hi = bf16(x), lo = bf16(x - hi) and a product is
// formed as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_16x16x32_bf16 (fp32 accumulate): ~2^-16 relative
// error per product instead of 2^-8, at 3/16 of the exact-f32 MFMA cost.  Residuals, biases and both LayerNorms
// stay fp32.
// =====================================================================================================================
namespace {

// Wp{hi,lo}[nt][ks][lane][8] = split(W[nt*16 + (lane&15)][ks*32 + (lane>>4)*8 + j])   (0 outside [N,K])
__global__ __launch_bounds__(256) void pack_w_split_kernel(const float* __restrict__ W, int N, int K,
                                                           unsigned short* __restrict__ Whi, unsigned short* __restrict__ Wlo) {
    const int KS = (K + 31) / 32, NTt = (N + 15) / 16;
    const size_t total = (size_t)NTt * KS * 64;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        size_t r = i >> 6;
        const int ks = (int)(r % KS);
        const int nt = (int)(r / KS);
        const int n = nt * 16 + (lane & 15);
        const int k0 = ks * 32 + (lane >> 4) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = (n < N && k0 + j < K) ? W[(size_t)n * K + k0 + j] : 0.f;
            split_store(Whi, Wlo, (int)(i * 8 + j), x);
        }
    }
}

template <int TERMS>
__global__ __launch_bounds__(NTHR) void mha_tail_bf16_kernel(const float* __restrict__ o, int HK, const float* __restrict__ q, int B,
                                                             TailW w, float eps, float* __restrict__ out, int HKn,
                                                             float* __restrict__ qh_next) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    mg_tail::tail_bf16_body<TERMS, false>(smem_b, o, HK, q, B, w, eps, out, HKn, qh_next, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}

// the same with the K of `fc` split over the cluster (round 4; as mha_tail_c16_kernel): ranks of a tile are ADJACENT workgroups,
// each contracts a slice of K, the rank that arrives last adds the partial sums and finishes the tile; the next layer's w_qs is a
// launch of its own behind it (mha_proj_c16_kernel)
template <int TERMS>
__global__ __launch_bounds__(NTHR) void mha_tail_bf16_ks_kernel(const float* __restrict__ o, int HK, const float* __restrict__ q, int B,
                                                                TailW w, float eps, float* __restrict__ out, int cl,
                                                                float* __restrict__ xpart, int* __restrict__ xcnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    mg_tail::tail_bf16_body<TERMS, false>(smem_b, o, HK, q, B, w, eps, out, 0, nullptr, (int)blockIdx.x / cl, (int)blockIdx.x % cl, cl,
                                          xpart, xcnt);
}

// the same tail behind the folded attention (sq_mha_folded_bf16.hip): `c` = bf16 [B, HC] weighted bank rows per head, `fc` = the
// composed map fc . blockdiag(W_v) [300, HC], the next projection = the composed W_k^T W_q rows [HCn, 300]
__global__ __launch_bounds__(NTHR) void mha_tail_c16_kernel(const unsigned short* __restrict__ c, int HC, const float* __restrict__ q,
                                                            int B, TailW w, float eps, float* __restrict__ out, int HCn,
                                                            float* __restrict__ u_next, int cl, float* __restrict__ xpart,
                                                            int* __restrict__ xcnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    mg_tail::tail_bf16_body<1, false, true>(smem_b, reinterpret_cast<const float*>(c), HC, q, B, w, eps, out, HCn, u_next,
                                            (int)blockIdx.x / cl, (int)blockIdx.x % cl, cl, xpart, xcnt);
}

// u = M x + c for a 16-sample tile x (fp32 [B, 300], a layer's output) and the next layer's composed query map M (packed bf16,
// [HCn, 300]): a launch of its own behind the last-arriver form of the tail above.  `cl` workgroups per tile share M's column
// tiles; each wave streams its tiles' fragments as ONE stream (k-step d of the next tile is requested when k-step d of the
// current one is consumed).  11 KB of LDS, ~100 registers: shares a CU with anything.
__global__ __launch_bounds__(NTHR) void mha_proj_c16_kernel(const float* __restrict__ x, int B, const unsigned short* __restrict__ wq_h,
                                                            const float* __restrict__ bq, int HCn, float* __restrict__ u_next, int cl) {
    __shared__ __attribute__((aligned(16))) uint4 s_a[ROWS * SCD];
    constexpr int KSd = (D + 31) / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = (int)blockIdx.x / cl, rank = (int)blockIdx.x % cl, r0 = tile * ROWS;
    const int NTq = (HCn + 15) / 16;
    const int slots = (NTq + 7) / 8;
    const int per = (slots + cl - 1) / cl;
    const int s_lo = rank * per, s_hi = min(slots, s_lo + per);
    if (s_lo >= s_hi) return;
    const uint4* Wq = reinterpret_cast<const uint4*>(wq_h);
    auto wq_off = [&](int slot) {
        const int nt = wave + 8 * slot;
        return ((size_t)(nt < NTq ? nt : 0) * KSd) * 64 + lane;
    };
    uint4 rq[KSd];
    {
        const size_t o0 = wq_off(s_lo);
#pragma unroll
        for (int d = 0; d < KSd; ++d) rq[d] = Wq[o0 + (size_t)d * 64];
    }
    // x tile -> bf16 A image [16][SCD chunks of 8] (zero padded)
    for (int i = tid; i < ROWS * SCD; i += NTHR) {
        const int r = i / SCD, c8 = (i - r * SCD) * 8;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (r0 + r < B) {
            const float* xr = x + (size_t)(r0 + r) * D + c8;
            if (c8 + 4 <= D) a = *reinterpret_cast<const f32x4*>(xr);
            if (c8 + 8 <= D) b = *reinterpret_cast<const f32x4*>(xr + 4);
        }
        s_a[i] = make_uint4(f2bf2_t(a[0], a[1]), f2bf2_t(a[2], a[3]), f2bf2_t(b[0], b[1]), f2bf2_t(b[2], b[3]));
    }
    __syncthreads();
    bf16x8 aq[KSd];
#pragma unroll
    for (int d = 0; d < KSd; ++d) aq[d] = __builtin_bit_cast(bf16x8, s_a[(lane & 15) * SCD + (lane >> 4) + 4 * d]);
    const int crow = (lane >> 4) * 4, ccol = lane & 15;
    for (int slot = s_lo; slot < s_hi; ++slot) {
        const int nt = wave + 8 * slot, n = nt * 16 + ccol;
        const bool live = nt < NTq && n < HCn;
        const float bv = (live && bq) ? bq[n] : 0.f;
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
                if (gr < B) u_next[(size_t)gr * HCn + n] = a2[r] + bv;
            }
        }
    }
}

}  // namespace

extern "C" size_t mgnns_packed_bf16_weight_bytes(int N, int K) {      // ONE of the two (hi / lo) buffers
    return (size_t)((N + 15) / 16) * ((K + 31) / 32) * 64 * 16;
}

extern "C" int mgnns_pack_weight_bf16_split(const float* W, int N, int K, void* Whi, void* Wlo, mgnns_stream_t stream) {
    MG_REQUIRE(W && Whi && Wlo && N > 0 && K > 0, "mgnns_pack_weight_bf16_split: bad arguments");
    const size_t total = (size_t)((N + 15) / 16) * ((K + 31) / 32) * 64;
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_w_split_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, W, N, K,
                       reinterpret_cast<unsigned short*>(Whi), reinterpret_cast<unsigned short*>(Wlo));
    MG_CHECK_LAUNCH("mgnns_pack_weight_bf16_split");
    return 0;
}

#ifdef MG_TAIL_TRACE
extern "C" int mgnns_debug_tail_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mg_tail::g_tail_trace), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int mgnns_mha_tail_bf16_fwd(const float* o, int HK, const float* q, int B, int d_model, int terms,
                                       const void* const* packed /* fc_h,fc_l,w1_h,w1_l,w2_h,w2_l,wq_h,wq_l */,
                                       const float* fc_b, const float* ln1_gamma, const float* ln1_beta, const float* b1,
                                       const float* b2, const float* ln2_gamma, const float* ln2_beta, float eps, float* out,
                                       const float* bq_next, int HK_next, float* qh_next, int cluster, float* cluster_scratch,
                                       int* cluster_counters, mgnns_stream_t stream) {
    MG_REQUIRE(o && q && packed && fc_b && ln1_gamma && ln1_beta && b1 && b2 && ln2_gamma && ln2_beta && out,
               "mgnns_mha_tail_bf16_fwd: null pointer");
    MG_REQUIRE(cluster >= 0 && cluster <= 8, "mgnns_mha_tail_bf16_fwd: cluster=%d (0 = default, 1..8)", cluster);
    MG_REQUIRE((cluster_scratch != nullptr) == (cluster_counters != nullptr), "mgnns_mha_tail_bf16_fwd: cluster scratch and counters go together");
    MG_REQUIRE(!cluster_scratch || mg_aligned16(cluster_scratch), "mgnns_mha_tail_bf16_fwd: cluster scratch must be 16-byte aligned");
    MG_REQUIRE(d_model == D, "mgnns_mha_tail_bf16_fwd: d_model=%d unsupported (300 only)", d_model);
    MG_REQUIRE(terms == 1 || terms == 3, "mgnns_mha_tail_bf16_fwd: terms must be 1 (bf16) or 3 (split-bf16)");
    MG_REQUIRE(HK > 0 && HK % 32 == 0 && HK <= 2048, "mgnns_mha_tail_bf16_fwd: n_head*d_v=%d unsupported (multiple of 32, <= 2048)", HK);
    for (int i = 0; i < 6; ++i) MG_REQUIRE(packed[i], "mgnns_mha_tail_bf16_fwd: packed weight %d missing", i);
    MG_REQUIRE(!packed[6] || (packed[7] && qh_next && HK_next > 0), "mgnns_mha_tail_bf16_fwd: next-layer projection incomplete");
    if (B <= 0) return 0;
    TailW w;
    w.fc_h = (const unsigned short*)packed[0]; w.fc_l = (const unsigned short*)packed[1];
    w.w1_h = (const unsigned short*)packed[2]; w.w1_l = (const unsigned short*)packed[3];
    w.w2_h = (const unsigned short*)packed[4]; w.w2_l = (const unsigned short*)packed[5];
    w.wq_h = (const unsigned short*)packed[6]; w.wq_l = (const unsigned short*)packed[7];
    w.fc_b = fc_b; w.g1 = ln1_gamma; w.be1 = ln1_beta; w.b1 = b1; w.b2 = b2; w.g2 = ln2_gamma; w.be2 = ln2_beta; w.bq = bq_next;
    const int so = (HK >> 3) + 2;
    const size_t lds = (size_t)(2 * ROWS * so + 2 * ROWS * SCD) * 16 + 2 * (size_t)ROWS * SD * sizeof(float);
    MG_REQUIRE(lds <= 160 * 1024, "mgnns_mha_tail_bf16_fwd: needs %zu B of LDS", lds);
    MG_DYN_LDS(mha_tail_bf16_kernel<1>, 160 * 1024);
    MG_DYN_LDS(mha_tail_bf16_kernel<3>, 160 * 1024);
    // with a next-layer projection: a cluster of workgroups per 16-sample tile, each recomputing the front part and taking a
    // share of the projection (MGNNS_TAIL_CLUSTER overrides: 1 = none).  Four while the chip has CUs to spare; TWO from 256
    // samples on, where the forward is bound by CU time and 64 workgroups x 21 us cost more than the shorter chain returns
    // (B=256: 0.811-0.820 ms per forward with 2, 0.830 with 4; B=128: equal; B=64: 0.453-0.458 with 4, 0.464-0.467 with 2)
    // terms == 3 (round 5): the same K split on split-bf16 operands; it has no projection launch of its own -- the caller leaves the
    // next layer's w_qs to an exact-fp32 GEMM behind this call (packed[6] == NULL)
    if (cluster_scratch && cluster != 1 && (terms == 1 || !packed[6])) {
        // K-split form (round 4): the ranks of a tile split fc's K = n_head * d_v, the last arriver finishes the tile; the next
        // layer's w_qs as a launch of its own on 8 workgroups per tile (csrc: mha_proj_c16_kernel -- x -> bf16, one MFMA chain
        // per column tile, exactly what the tail's own projection phase does)
        int cl = cluster ? cluster : 4;
        if (cl > HK / 32) cl = HK / 32;                  // every rank needs a k-step of its own
        const unsigned tiles = (unsigned)((B + ROWS - 1) / ROWS);
        TailW w2 = w;
        w2.wq_h = w2.wq_l = nullptr;
        if (terms == 3) {
            MG_DYN_LDS(mha_tail_bf16_ks_kernel<3>, 160 * 1024);
            hipLaunchKernelGGL(mha_tail_bf16_ks_kernel<3>, dim3(tiles * cl), dim3(NTHR), lds, (hipStream_t)stream, o, HK, q, B, w2, eps,
                               out, cl, cluster_scratch, cluster_counters);
        } else {
            MG_DYN_LDS(mha_tail_bf16_ks_kernel<1>, 160 * 1024);
            hipLaunchKernelGGL(mha_tail_bf16_ks_kernel<1>, dim3(tiles * cl), dim3(NTHR), lds, (hipStream_t)stream, o, HK, q, B, w2, eps,
                               out, cl, cluster_scratch, cluster_counters);
        }
        MG_CHECK_LAUNCH("mgnns_mha_tail_bf16_fwd(K split)");
        if (packed[6]) {
            const int pcl = 8;
            hipLaunchKernelGGL(mha_proj_c16_kernel, dim3(tiles * pcl), dim3(NTHR), 0, (hipStream_t)stream, (const float*)out, B,
                               (const unsigned short*)packed[6], bq_next, HK_next, qh_next, pcl);
            MG_CHECK_LAUNCH("mgnns_mha_tail_bf16_fwd(projection)");
        }
        return 0;
    }
    int cl = packed[6] ? (B >= 256 ? 2 : 4) : 1;
    if (const int e = mg_env_int("MGNNS_TAIL_CLUSTER", 0, 1)) cl = packed[6] ? e : 1;
    if (cluster) cl = packed[6] ? cluster : 1;
    if (cl < 1) cl = 1;
    if (cl > 8) cl = 8;
    dim3 grid((B + ROWS - 1) / ROWS, cl);
    if (terms == 3)
        hipLaunchKernelGGL(mha_tail_bf16_kernel<3>, grid, dim3(NTHR), lds, (hipStream_t)stream, o, HK, q, B, w, eps, out, HK_next, qh_next);
    else
        hipLaunchKernelGGL(mha_tail_bf16_kernel<1>, grid, dim3(NTHR), lds, (hipStream_t)stream, o, HK, q, B, w, eps, out, HK_next, qh_next);
    MG_CHECK_LAUNCH("mgnns_mha_tail_bf16_fwd");
    return 0;
}

extern "C" size_t mgnns_mha_tail_c16_scratch_floats(int B, int cluster) {
    return (size_t)((B + ROWS - 1) / ROWS) * (size_t)(cluster < 1 ? 1 : cluster) * (8 * 3 * 64 * 4);
}

extern "C" int mgnns_mha_tail_c16_fwd(const void* c_bf16, int HC, const float* q, int B, int d_model,
                                      const void* const* packed /* fc_h,-,w1_h,-,w2_h,-,wq_h,- (hi parts only are read) */,
                                      const float* fc_b, const float* ln1_gamma, const float* ln1_beta, const float* b1,
                                      const float* b2, const float* ln2_gamma, const float* ln2_beta, float eps, float* out,
                                      const float* bq_next, int HC_next, float* u_next, int cluster, float* cluster_scratch,
                                      int* cluster_counters, mgnns_stream_t stream) {
    MG_REQUIRE(c_bf16 && q && packed && fc_b && ln1_gamma && ln1_beta && b1 && b2 && ln2_gamma && ln2_beta && out,
               "mgnns_mha_tail_c16_fwd: null pointer");
    MG_REQUIRE(d_model == D, "mgnns_mha_tail_c16_fwd: d_model=%d unsupported (300 only)", d_model);
    MG_REQUIRE(HC > 0 && HC % 32 == 0 && HC <= 2560, "mgnns_mha_tail_c16_fwd: n_head*d_model=%d unsupported (multiple of 32, <= 2560)", HC);
    MG_REQUIRE(mg_aligned16(c_bf16), "mgnns_mha_tail_c16_fwd: c must be 16-byte aligned");
    for (int i = 0; i < 6; i += 2) MG_REQUIRE(packed[i], "mgnns_mha_tail_c16_fwd: packed weight %d missing", i);
    MG_REQUIRE(!packed[6] || (u_next && HC_next > 0), "mgnns_mha_tail_c16_fwd: next-layer projection incomplete");
    MG_REQUIRE(cluster >= 0 && cluster <= 8, "mgnns_mha_tail_c16_fwd: cluster=%d (0 = default, 1..8)", cluster);
    MG_REQUIRE((cluster_scratch != nullptr) == (cluster_counters != nullptr), "mgnns_mha_tail_c16_fwd: cluster scratch and counters go together");
    MG_REQUIRE(!cluster_scratch || mg_aligned16(cluster_scratch), "mgnns_mha_tail_c16_fwd: cluster scratch must be 16-byte aligned");
    if (B <= 0) return 0;
    TailW w;
    w.fc_h = (const unsigned short*)packed[0]; w.fc_l = nullptr;
    w.w1_h = (const unsigned short*)packed[2]; w.w1_l = nullptr;
    w.w2_h = (const unsigned short*)packed[4]; w.w2_l = nullptr;
    w.wq_h = (const unsigned short*)packed[6]; w.wq_l = nullptr;
    w.fc_b = fc_b; w.g1 = ln1_gamma; w.be1 = ln1_beta; w.b1 = b1; w.b2 = b2; w.g2 = ln2_gamma; w.be2 = ln2_beta; w.bq = bq_next;
    const int so = (HC >> 3) + 2;
    const size_t lds = (size_t)(ROWS * so + 2 * ROWS * SCD) * 16 + 2 * (size_t)ROWS * SD * sizeof(float);
    MG_REQUIRE(lds <= 160 * 1024, "mgnns_mha_tail_c16_fwd: needs %zu B of LDS", lds);
    MG_DYN_LDS(mha_tail_c16_kernel, 160 * 1024);
    // A cluster of workgroups per 16-sample tile.  With exchange buffers the ranks split the K of the first product (the composed
    // output map); without buffers every rank repeats the front part and takes a share of the next composed query map's columns
    // (or, without a next map and cluster == 0, one workgroup per tile).
    // (measured at B = 256, two forwards in flight: 442 k samples/s with four ranks, 435 k with two; one at a time 0.624 / 0.643 ms)
    int cl = cluster ? cluster : (cluster_scratch ? 4 : (packed[6] ? (B >= 256 ? 2 : 4) : 1));
    if (!cluster) {
        if (const int e = mg_env_int("MGNNS_TAIL_CLUSTER", 0, 1)) cl = e;
    }
    if (cl < 1) cl = 1;
    if (cl > 8) cl = 8;
    if (!packed[6] && !cluster_scratch) cl = 1;            // nothing to share
    const unsigned tiles = (unsigned)((B + ROWS - 1) / ROWS);
    // With exchange buffers and more than one rank: the LAST rank of a tile to arrive adds the partial sums and runs the rest of
    // the tail alone (nobody waits; the LayerNorm / FFN chain runs once per tile), and the next layer's composed query map is a
    // launch of its own on 8 workgroups per tile.
    const bool last_arriver = cluster_scratch && cl > 1;
    if (last_arriver) w.wq_h = nullptr;
    hipLaunchKernelGGL(mha_tail_c16_kernel, dim3(tiles * cl), dim3(NTHR), lds, (hipStream_t)stream,
                       static_cast<const unsigned short*>(c_bf16), HC, q, B, w, eps, out, HC_next, u_next, cl,
                       last_arriver ? cluster_scratch : (float*)nullptr, last_arriver ? cluster_counters : (int*)nullptr);
    MG_CHECK_LAUNCH("mgnns_mha_tail_c16_fwd");
    if (last_arriver && packed[6]) {
        const int pcl = 8;
        hipLaunchKernelGGL(mha_proj_c16_kernel, dim3(tiles * pcl), dim3(NTHR), 0, (hipStream_t)stream, (const float*)out, B,
                           (const unsigned short*)packed[6], bq_next, HC_next, u_next, pcl);
        MG_CHECK_LAUNCH("mgnns_mha_tail_c16_fwd(projection)");
    }
    return 0;
}
