// The label-attention tail of an image channel in ONE launch (MODEL:88-133 `Attention.forward` minus its w_q, plus
// MODEL:477-479 / 504-506): from the read-out x = pooled . G^T  [B, C] to the channel feature [B, 300]
//   K = w_k x + b_k, V = w_v x + b_v                              [B, 300] each
//   o[b,l,h,:] = softmax_d(Q[l,h,:] * K[b,h,:] / sqrt(dh)) * V[b,h,:]      (element-wise "attention", l < NLQ label rows)
//   y[b,l,:]  = linear_5(fc(o[b,l,:]))  = Wc o + bc,  Wc = W5 Wfc  [N5, 300]  (no non-linearity between the two maps:
//                                                          they are composed ONCE per weight version on the host side)
//   out[b,:]  = x_linear(flatten_l y[b,l,:])                      [B, 300]
// The reference runs this as 6 GEMMs + ~10 element-wise kernels and a Python loop over the batch (MODEL:114-115); as
// separate launches here it was 8 small launches that each had to wait for a free CU behind the chip-filling attention
// and memory-bank kernels of the other streams (200-290 us on the critical path of a 1 ms forward,
// tools/graph_timeline.py).  One workgroup owns 16 samples through the chain: activations in LDS, weights streamed from
// L2 in the fragment-major fp32 layout of mgnns_pack_weight_f32, every contraction on the exact-f32 MFMA.
#include "common.hpp"
#include "tile_f32.hpp"

namespace {

constexpr int LT_THR = 512;
constexpr int LT_ROWS = 16;
constexpr int LT_MAXKQ = 20;                 // hid <= 320: k-quads of the composed map held in registers
constexpr int LT_MAXH = 8;                   // heads of the label attention (the reference hard-codes 5, MODEL:312-313)

__host__ __device__ inline int lt_stride(int k) {          // LDS row stride: >= k rounded to 16, == 2 (mod 32)
    const int kp = (k + 15) / 16 * 16;
    return kp + ((34 - (kp % 32)) % 32);
}

constexpr int LT_KCH = 512;                  // read-out K chunk staged at a time (aliases the flatten buffer)

__global__ __launch_bounds__(LT_THR) void label_tail_kernel(const float* __restrict__ x, int B, int C,
                                                            const float* __restrict__ pooled, int n_parts, int KP,
                                                            const float* __restrict__ g_wp,
                                                            const float* __restrict__ Q, int NLQ, int n_heads, int dh,
                                                            const float* __restrict__ wk_wp, const float* __restrict__ bk,
                                                            const float* __restrict__ wv_wp, const float* __restrict__ bv,
                                                            const float* __restrict__ wc_wp, const float* __restrict__ bc, int N5,
                                                            const float* __restrict__ xl_wp, const float* __restrict__ bxl, int NO,
                                                            float* __restrict__ out, const float* __restrict__ wq_wp,
                                                            const float* __restrict__ bq, int HKn, float* __restrict__ qh_next) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int hid = n_heads * dh;
    const int sx = lt_stride(C), sh = lt_stride(hid), sf = lt_stride(NLQ * N5);
    float* s_x = smem;                       // [16][sx]
    float* s_k = s_x + LT_ROWS * sx;         // [16][sh]
    float* s_v = s_k + LT_ROWS * sh;
    float* s_o = s_v + LT_ROWS * sh;
    float* s_f = s_o + LT_ROWS * sh;         // [16][sf]  flatten_l y[b,l,:]
    float* s_q = s_f + LT_ROWS * sf;         // [NLQ][hid] the projected label query (read 70 x per wave in the label loop)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * LT_ROWS;
    const int crow = (lane >> 4) * 4, ccol = lane & 15;
    const int NTh = (hid + 15) / 16, NT5 = (N5 + 15) / 16, NTo = (NO + 15) / 16;
    const int KQh = (hid + 15) / 16;

    // the composed map's fragments of this wave's column tile stay in registers across the NLQ label rows
    f32x4 wc[LT_MAXKQ];
    if (wave < NT5) {
#pragma unroll
        for (int kq = 0; kq < LT_MAXKQ; ++kq)
            wc[kq] = kq < KQh ? reinterpret_cast<const f32x4*>(wc_wp)[((size_t)wave * KQh + kq) * 64 + lane] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // ---- zero every buffer's padding; x = the read-out, either given or computed here ---------------------------------------
    for (int i = tid; i < LT_ROWS * sx + 3 * LT_ROWS * sh + LT_ROWS * sf; i += LT_THR) s_x[i] = 0.f;
    for (int i = tid; i < NLQ * hid; i += LT_THR) s_q[i] = Q[i];
    __syncthreads();
    if (g_wp) {
        // x = max_parts(pooled) . G^T (MODEL:454-455 + 474): K = 2048 does not fit LDS next to everything else, so it is
        // walked in chunks of 512 staged into the (still unused) flatten buffer; G streams from L2 in packed form
        const int NTc = (C + 15) / 16, KQ = (KP + 15) / 16;
        const int sp = lt_stride(LT_KCH);
        float* s_p = s_f;
        for (int t0 = 0; t0 * 8 < NTc; t0 += 3) {
            f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            for (int k0 = 0; k0 < KP; k0 += LT_KCH) {
                const int kn = min(LT_KCH, KP - k0);
                for (int i = tid; i < LT_ROWS * (LT_KCH / 4); i += LT_THR) {
                    const int r = i / (LT_KCH / 4), c4 = (i - r * (LT_KCH / 4)) * 4;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (r0 + r < B && c4 < kn) {            // KP % 4 == 0 (launcher)
                        const float* src = pooled + ((size_t)(r0 + r) * n_parts) * KP + k0 + c4;
                        v = *reinterpret_cast<const f32x4*>(src);
                        for (int pt = 1; pt < n_parts; ++pt) {
                            const f32x4 u = *reinterpret_cast<const f32x4*>(src + (size_t)pt * KP);
                            v = f32x4{fmaxf(v.x, u.x), fmaxf(v.y, u.y), fmaxf(v.z, u.z), fmaxf(v.w, u.w)};
                        }
                    }
                    float* d = s_p + r * sp + c4;           // sp is even: 8-byte aligned
                    *reinterpret_cast<float2*>(d) = float2{v.x, v.y};
                    *reinterpret_cast<float2*>(d + 2) = float2{v.z, v.w};
                }
                __syncthreads();
                mg_tile_gemm_f32_chunk<3>(acc, s_p, sp, k0 / 16, (k0 + kn + 15) / 16, KQ, g_wp, NTc, wave, lane, t0);
                __syncthreads();
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int nt = wave + 8 * (t0 + t), n = nt * 16 + ccol;
                if (nt < NTc && n < C) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) s_x[(crow + r) * sx + n] = acc[t][r];
                }
            }
        }
        for (int i = tid; i < LT_ROWS * sf; i += LT_THR) s_f[i] = 0.f;      // the flatten buffer's padding again
    } else {
        for (int i = tid; i < LT_ROWS * C; i += LT_THR) {
            const int r = i / C, c = i - r * C;
            if (r0 + r < B) s_x[r * sx + c] = x[(size_t)(r0 + r) * C + c];
        }
    }
    __syncthreads();

    // ---- K, V -----------------------------------------------------------------------------------------------------
    for (int t0 = 0; t0 * 8 < NTh; t0 += 3) {
        f32x4 ak[3], av[3];
        mg_tile_gemm_f32<3>(ak, s_x, sx, C, wk_wp, NTh, wave, lane, t0);
        mg_tile_gemm_f32<3>(av, s_x, sx, C, wv_wp, NTh, wave, lane, t0);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int nt = wave + 8 * (t0 + t), n = nt * 16 + ccol;
            if (nt < NTh && n < hid) {
                const float b0 = bk[n], b1 = bv[n];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s_k[(crow + r) * sh + n] = ak[t][r] + b0;
                    s_v[(crow + r) * sh + n] = av[t][r] + b1;
                }
            }
        }
    }
    __syncthreads();

    // ---- per label row l: element-wise attention into s_o, composed map into its slot of s_f --------------------------
    const float inv_scale = 1.0f / sqrtf((float)dh);
    for (int l = 0; l < NLQ; ++l) {
        // wave w owns sample rows 2w, 2w+1; a lane holds dim `lane` of EVERY head (the independent reductions interleave)
#pragma unroll 1
        for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * wave + rr;
            const bool on = lane < dh;
            float e[LT_MAXH], vv[LT_MAXH];
#pragma unroll
            for (int h = 0; h < LT_MAXH; ++h) {
                const int c = (h < n_heads ? h : 0) * dh + (on ? lane : 0);
                e[h] = (on && h < n_heads) ? s_q[l * hid + c] * s_k[r * sh + c] * inv_scale : -INFINITY;
                vv[h] = s_v[r * sh + c];
            }
#pragma unroll
            for (int h = 0; h < LT_MAXH; ++h) {
                if (h < n_heads) {                                  // wave-uniform
                    const float m = wave_max_dpp(e[h]);
                    const float p = on ? expf(e[h] - m) : 0.f;
                    const float z = wave_sum_dpp(p);
                    if (on) s_o[r * sh + h * dh + lane] = (p / z) * vv[h];
                }
            }
        }
        __syncthreads();
        if (wave < NT5) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* ap = s_o + (lane & 15) * sh + (lane >> 4);
#pragma unroll
            for (int kq = 0; kq < LT_MAXKQ; ++kq) {
                if (kq < KQh) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[(4 * kq + j) * 4], wc[kq][j], acc, 0, 0, 0);
                }
            }
            const int n = wave * 16 + ccol;
            if (n < N5) {
                const float b0 = bc[n];
#pragma unroll
                for (int r = 0; r < 4; ++r) s_f[(crow + r) * sf + l * N5 + n] = acc[r] + b0;
            }
        }
        __syncthreads();
    }

    // ---- out = x_linear(flat) ------------------------------------------------------------------------------------------
    for (int t0 = 0; t0 * 8 < NTo; t0 += 3) {
        f32x4 acc[3];
        mg_tile_gemm_f32<3>(acc, s_f, sf, NLQ * N5, xl_wp, NTo, wave, lane, t0);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int nt = wave + 8 * (t0 + t), n = nt * 16 + ccol;
            if (nt < NTo && n < NO) {
                const float b0 = bxl[n];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gr = r0 + crow + r;
                    const float v = acc[t][r] + b0;
                    if (gr < B) out[(size_t)gr * NO + n] = v;
                    if (wq_wp) s_o[(crow + r) * sh + n] = v;          // NO == hid when the projection is asked for (launcher)
                }
            }
        }
    }
    // ---- the query projection of the fusion stack this feature feeds: qh = w_qs(out) + b (submodules.py:63-66) ----------------
    if (wq_wp) {
        __syncthreads();
        const int NTq = (HKn + 15) / 16;
        for (int t0 = 0; t0 * 8 < NTq; t0 += 3) {
            f32x4 acc[3];
            mg_tile_gemm_f32<3>(acc, s_o, sh, NO, wq_wp, NTq, wave, lane, t0);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int nt = wave + 8 * (t0 + t), n = nt * 16 + ccol;
                if (nt < NTq && n < HKn) {
                    const float b0 = bq ? bq[n] : 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int gr = r0 + crow + r;
                        if (gr < B) qh_next[(size_t)gr * HKn + n] = acc[t][r] + b0;
                    }
                }
            }
        }
    }
}

}  // namespace

// The shape limits of mgnns_label_tail_fwd as a predicate (the host side routes unsupported shapes to the operator chain
// instead of running into the launcher's argument errors).  K_pool = 0: the read-out x is passed in.
extern "C" int mgnns_label_tail_supported(int C, int NLQ, int n_heads, int dh, int N5, int n_out, int K_pool, int with_next_q) {
    if (!(C > 0 && NLQ > 0 && n_heads > 0 && dh > 0 && dh <= 64 && N5 > 0 && n_out > 0)) return 0;
    if (K_pool && K_pool % 4) return 0;
    const int hid = n_heads * dh;
    if (!(hid <= 16 * LT_MAXKQ && n_heads <= LT_MAXH && N5 <= 128)) return 0;
    if (with_next_q && n_out != hid) return 0;
    if (K_pool && lt_stride(NLQ * N5) < lt_stride(LT_KCH)) return 0;
    const size_t lds = ((size_t)LT_ROWS * (lt_stride(C) + 3 * lt_stride(hid) + lt_stride(NLQ * N5)) + (size_t)NLQ * hid) * sizeof(float);
    return lds <= 160 * 1024 ? 1 : 0;
}

extern "C" int mgnns_label_tail_fwd(const float* x, int B, int C, const float* pooled, int n_parts, int K_pool,
                                    const float* g_wp, const float* Q, int NLQ, int n_heads, int dh,
                                    const float* wk_wp, const float* bk, const float* wv_wp, const float* bv,
                                    const float* wc_wp, const float* bc, int N5, const float* xl_wp, const float* bxl,
                                    int n_out, float* out, const float* wq_next_wp, const float* bq_next, int HK_next,
                                    float* qh_next, mgnns_stream_t stream) {
    MG_REQUIRE(B >= 0 && C > 0 && NLQ > 0 && n_heads > 0 && dh > 0 && dh <= 64 && N5 > 0 && n_out > 0,
               "mgnns_label_tail_fwd: bad dims B=%d C=%d NLQ=%d heads=%d dh=%d N5=%d out=%d", B, C, NLQ, n_heads, dh, N5, n_out);
    if (B == 0) return 0;
    MG_REQUIRE(Q && wk_wp && bk && wv_wp && bv && wc_wp && bc && xl_wp && bxl && out, "mgnns_label_tail_fwd: null pointer");
    MG_REQUIRE((x != nullptr) != (g_wp != nullptr), "mgnns_label_tail_fwd: pass EITHER the read-out x OR pooled + packed G");
    if (g_wp) {
        MG_REQUIRE(pooled && n_parts >= 1 && K_pool > 0 && K_pool % 4 == 0 && mg_aligned16(pooled),
                   "mgnns_label_tail_fwd: pooled [B,%d,%d] must be 16-byte aligned with K %% 4 == 0", n_parts, K_pool);
    }
    const int hid = n_heads * dh;
    MG_REQUIRE(hid <= 16 * LT_MAXKQ && n_heads <= LT_MAXH, "mgnns_label_tail_fwd: hidden width %d / %d heads unsupported (<= %d, <= %d)",
               hid, n_heads, 16 * LT_MAXKQ, LT_MAXH);
    MG_REQUIRE(N5 <= 128, "mgnns_label_tail_fwd: linear_5 width %d unsupported (<= 128)", N5);
    MG_REQUIRE(!wq_next_wp || (qh_next && HK_next > 0 && n_out == hid),
               "mgnns_label_tail_fwd: the query projection needs qh_next, HK_next and n_out == hidden width");
    const int sfl = lt_stride(NLQ * N5) > lt_stride(LT_KCH) || !g_wp ? lt_stride(NLQ * N5) : lt_stride(LT_KCH);
    MG_REQUIRE(!g_wp || lt_stride(NLQ * N5) >= lt_stride(LT_KCH), "mgnns_label_tail_fwd: NLQ*N5=%d too small to stage the read-out (>= %d)",
               NLQ * N5, LT_KCH);
    (void)sfl;
    const size_t lds = ((size_t)LT_ROWS * (lt_stride(C) + 3 * lt_stride(hid) + lt_stride(NLQ * N5)) + (size_t)NLQ * hid) * sizeof(float);
    MG_REQUIRE(lds <= 160 * 1024, "mgnns_label_tail_fwd: C=%d, NLQ*N5=%d need %zu B of LDS (> 160 KiB)", C, NLQ * N5, lds);
    MG_DYN_LDS(label_tail_kernel, 160 * 1024);
    hipLaunchKernelGGL(label_tail_kernel, dim3((B + LT_ROWS - 1) / LT_ROWS), dim3(LT_THR), lds, (hipStream_t)stream, x, B, C,
                       pooled, n_parts, K_pool, g_wp, Q, NLQ, n_heads, dh, wk_wp, bk, wv_wp, bv, wc_wp, bc, N5, xl_wp, bxl, n_out,
                       out, wq_next_wp, bq_next, HK_next, qh_next);
    MG_CHECK_LAUNCH("mgnns_label_tail_fwd");
    return 0;
}

// =====================================================================================================================
// bf16 precision mode: the WHOLE channel tail behind the memory-bank kernel in one launch, read-out included --
//   x = max_parts(pooled) . G^T,  K/V,  element-wise attention,  Wc,  x_linear,  the next stack's w_qs
// -- every contraction on v_mfma_f32_16x16x32_bf16 (bf16 operands, fp32 accumulation; softmax, biases and the attention
// products in fp32).  On the exact-f32 MFMA the read-out alone is 40 us of matrix pipe on the 16 CUs a 256-sample batch
// gives this kernel (which is why the fp32 mode keeps it as a separate, chip-wide GEMM launch); in bf16 it is 2 us and the
// kernel is bound by streaming ~3 MB of packed weights per workgroup from L2.  One launch instead of three matters because
// every launch of this chain waits for a free CU behind the chip-filling attention cores of the other streams
// (tools/graph_timeline.py: 200-270 us for the three launches, on the critical path of the two image->text stacks).
// =====================================================================================================================
#include "tile_bf16.hpp"

#ifdef MG_LT_TRACE
__device__ unsigned long long g_lt_trace[16];
#define LT_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_lt_trace[i] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int mgnns_debug_lt_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lt_trace), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1;
}
#else
#define LT_STAMP(i) do { } while (0)
#endif

namespace {

constexpr int LB_MAXKS = 10;                 // hid <= 320: k-steps of the composed map held in registers

struct LabelW {             // packed hi / lo buffers (lo unused with TERMS == 1) + fp32 vectors
    const unsigned short *g_h, *g_l, *wk_h, *wk_l, *wv_h, *wv_l, *wc_h, *wc_l, *xl_h, *xl_l, *wq_h, *wq_l;
    const float *bk, *bv, *bc, *bxl, *bq;
};

typedef int lt_i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk2(float a, float b) { return f2bf_t(a) | (unsigned)f2bf_t(b) << 16; }

// TERMS = 1: plain bf16 operands (+1e-2 on the logits of the B=256 batch: the label-attention feature is the QUERY of two
// fusion stacks, every bf16 stage of this chain costs ~1e-2 there).  TERMS = 3 (default): split-bf16 operands, three MFMAs
// per product, fp32-class accuracy at twice the weight bytes -- still one launch.
//
// CL = 4 (split-bf16 only): a CLUSTER of four workgroups shares a 16-sample tile.  The two wide phases are divided --
// rank r contracts the r-th quarter of the read-out's K and takes a quarter of the next query's column tiles -- and the
// narrow middle (K/V, label loop, x_linear: 44 % of the single-workgroup time) is recomputed by every rank.  The read-out
// partials cross through global memory without fences: system-scope write-through stores, vmcnt(0), ONE relaxed agent-scope
// arrival count per workgroup, cache-bypassing loads; every rank adds the four partials in the same order, so the ranks hold
// identical x.  A departure count lets the last rank to finish reading re-arm both counters for the next launch.
template <int TERMS, int CL>
__global__ __launch_bounds__(LT_THR) void label_tail_bf16_kernel(const float* __restrict__ pooled, int B, int n_parts, int KP, int C,
                                                                 const float* __restrict__ Q, int NLQ, int n_heads, int dh,
                                                                 LabelW w, int N5, int NO, float* __restrict__ out, int HKn,
                                                                 float* __restrict__ qh_next, float* __restrict__ xpart,
                                                                 int* __restrict__ counters, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    constexpr int NH = TERMS == 3 ? 4 : 1;                              // passes over the read-out's K (LDS holds K / NH of pooled)
    constexpr int LO = TERMS == 3 ? 1 : 0;
    const int hid = n_heads * dh;
    const int KSp = KP / 32, KSpp = KSp / NH, KSx = (C + 31) / 32, KSh = (hid + 31) / 32, KSf = (NLQ * N5 + 31) / 32;
    const int sp = 4 * KSpp + 2, sxc = 4 * KSx + 2, shc = 4 * KSh + 2, sfc = 4 * KSf + 2;     // chunk strides, == 2 (mod 4)
    const int shf = lt_stride(hid);                                                             // fp32 stride of K / V
    const int spf = sp > sfc ? sp : sfc;
    uint4* s_ph = reinterpret_cast<uint4*>(smem_b);                  // [16][sp]   pooled K-part hi; later the flatten buffer hi
    uint4* s_pl = s_ph + LT_ROWS * spf;                              //            ... lo (TERMS == 3)
    uint4* s_xh = s_pl + LO * LT_ROWS * spf;                         // [16][sxc]  read-out
    uint4* s_xl = s_xh + LT_ROWS * sxc;
    uint4* s_oh = s_xl + LO * LT_ROWS * sxc;                         // [16][shc]  attention output of one label row / out
    uint4* s_ol = s_oh + LT_ROWS * shc;
    float* s_k = reinterpret_cast<float*>(s_ol + LO * LT_ROWS * shc);   // [16][shf]  fp32
    float* s_v = s_k + LT_ROWS * shf;
    float* s_q = s_v + LT_ROWS * shf;                                   // [NLQ][hid] the projected label query (read 70 x per wave)
    uint4* s_fh = s_ph;
    uint4* s_fl = s_pl;
    if (!LO) { s_pl = s_ph; s_xl = s_xh; s_ol = s_oh; s_fl = s_fh; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = (int)blockIdx.x / CL, crank = (int)blockIdx.x % CL;      // a tile's ranks are adjacent in dispatch order
    const int r0 = tile * LT_ROWS;
    const int crow = (lane >> 4) * 4, ccol = lane & 15;
    const int NTc = (C + 15) / 16, NTh = (hid + 15) / 16, NT5 = (N5 + 15) / 16, NTo = (NO + 15) / 16;
    unsigned short* xh16 = reinterpret_cast<unsigned short*>(s_xh);
    unsigned short* xl16 = reinterpret_cast<unsigned short*>(s_xl);
    unsigned short* oh16 = reinterpret_cast<unsigned short*>(s_oh);
    unsigned short* ol16 = reinterpret_cast<unsigned short*>(s_ol);
    unsigned short* fh16 = reinterpret_cast<unsigned short*>(s_fh);
    unsigned short* fl16 = reinterpret_cast<unsigned short*>(s_fl);
    auto put = [&](unsigned short* h, unsigned short* l, int idx, float v) {
        if (LO) split_store(h, l, idx, v); else h[idx] = f2bf_t(v);
    };

    // split-bf16 streams twice the bytes per k-step: six k-steps in flight instead of the tail kernel's three (this kernel has
    // the registers: its accumulators are 12 VGPRs)
    LT_STAMP(0);
#ifndef MG_LT_TOUCH
#define MG_LT_TOUCH 0      // measured (NOTES_r05): touching the weights into L2 up front costs more (28 k cycles of scattered loads) than it saves
#endif
#ifndef MG_LT_PF3
#define MG_LT_PF3 4
#endif
    WRing<3, TERMS, (TERMS == 3 ? MG_LT_PF3 : 10)> ring;
    ring_prime(ring, KSpp, w.g_h, w.g_l, NTc, wave, lane, 0, KSp, CL > 1 ? crank * KSpp : 0);     // G flies through the staging of pooled
#if MG_LT_TOUCH
    // Every weight this workgroup will stream is TOUCHED now (one dword per 128-byte line, results never read): the chain of small
    // GEMMs below is bound by the latency of its weight fragments, and on a workgroup's XCD they are cold -- the launch has a few
    // workgroups per XCD and each reads its 2.4 MB once -- so every k-step waited for an HBM round trip (KV: 1300 cycles per k-step
    // of nine MFMAs).  Touched up front the lines sit in the XCD's L2 when the rings ask for them.  (asm loads: the compiler neither
    // drops them nor waits for them; its own counted waits only get stricter with more loads in flight.  They all land in ONE
    // register that stays allocated until the end of the kernel -- `touch_sink` is an in/out operand of every load and of the
    // final wait: a dead destination would be handed to another value and overwritten when the load lands.)
    int touch_sink = 0;
    {
        auto touch = [&](const unsigned short* base, int nt_n, int ks_total, int ks_lo, int ks_n, int nt_lo = 0) {
            if (!base || nt_n <= 0) return;
            base += ((size_t)nt_lo * ks_total) << 9;                   // (bf16 elements: 512 per fragment)
            const int lines = nt_n * ks_n * 8;                         // 128-byte lines (a fragment = 1 KiB = 8 lines)
            for (int i = tid; i < lines; i += LT_THR) {
                const int nt = i / (ks_n * 8), rem = i - nt * (ks_n * 8);
                const char* p = reinterpret_cast<const char*>(base) + ((size_t)(nt * ks_total + ks_lo + (rem >> 3)) << 10) + ((rem & 7) << 7);
                asm volatile("global_load_dword %0, %1, off" : "+v"(touch_sink) : "v"(p) : "memory");
            }
        };
        const int NTq_ = (HKn + 15) / 16;
        touch(w.g_h, NTc, KSp, CL > 1 ? crank * KSpp : 0, CL > 1 ? KSpp : KSp);
        if (LO) touch(w.g_l, NTc, KSp, CL > 1 ? crank * KSpp : 0, CL > 1 ? KSpp : KSp);
        touch(w.wk_h, NTh, KSx, 0, KSx);
        if (LO) touch(w.wk_l, NTh, KSx, 0, KSx);
        touch(w.wv_h, NTh, KSx, 0, KSx);
        if (LO) touch(w.wv_l, NTh, KSx, 0, KSx);
        touch(w.wc_h, NT5, KSh, 0, KSh);
        if (LO) touch(w.wc_l, NT5, KSh, 0, KSh);
        touch(w.xl_h, NTo, KSf, 0, KSf);
        if (LO) touch(w.xl_l, NTo, KSf, 0, KSf);
        // (the next query's column tiles of THIS rank only: slots q_lo .. q_hi of eight tiles, computed again below)
        const int qs_ = (NTq_ + 7) / 8, qp_ = (qs_ + CL - 1) / CL;
        const int qt_lo = crank * qp_ * 8, qt_hi = (crank * qp_ + qp_) * 8 < NTq_ ? (crank * qp_ + qp_) * 8 : NTq_;
        touch(w.wq_h, qt_hi - qt_lo, KSh, 0, KSh, qt_lo);
        if (LO) touch(w.wq_l, qt_hi - qt_lo, KSh, 0, KSh, qt_lo);
    }
#endif
    for (int i = tid; i < (1 + LO) * LT_ROWS * (sxc + shc); i += LT_THR) s_xh[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < 2 * LT_ROWS * shf; i += LT_THR) s_k[i] = 0.f;
    for (int i = tid; i < NLQ * hid; i += LT_THR) s_q[i] = Q[i];

    // ---- x = max_parts(pooled) . G^T (MODEL:454-455 + 474), K walked in NH parts ----------------------------------------------
    f32x4 acc[3];
    LT_STAMP(1);
    static_assert(CL == 1 || CL == NH, "a cluster rank owns one K part of the read-out");
    for (int hpart = CL > 1 ? crank : 0; hpart < (CL > 1 ? crank + 1 : NH); ++hpart) {
        const int kq8 = KP / NH / 8;
        for (int i = tid; i < LT_ROWS * kq8; i += LT_THR) {
            const int r = i / kq8, c8 = i - r * kq8;
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
            if (r0 + r < B) {
                const float* src = pooled + ((size_t)(r0 + r) * n_parts) * KP + hpart * (KP / NH) + 8 * c8;
                a = *reinterpret_cast<const f32x4*>(src);
                b = *reinterpret_cast<const f32x4*>(src + 4);
                for (int pt = 1; pt < n_parts; ++pt) {
                    const f32x4 u = *reinterpret_cast<const f32x4*>(src + (size_t)pt * KP);
                    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)pt * KP + 4);
                    a = f32x4{fmaxf(a.x, u.x), fmaxf(a.y, u.y), fmaxf(a.z, u.z), fmaxf(a.w, u.w)};
                    b = f32x4{fmaxf(b.x, v.x), fmaxf(b.y, v.y), fmaxf(b.z, v.z), fmaxf(b.w, v.w)};
                }
            }
            s_ph[r * sp + c8] = make_uint4(pk2(a.x, a.y), pk2(a.z, a.w), pk2(b.x, b.y), pk2(b.z, b.w));
            if (LO) {
                auto lo = [](float x) { return x - bf2f_t(f2bf_t(x)); };
                s_pl[r * sp + c8] = make_uint4(pk2(lo(a.x), lo(a.y)), pk2(lo(a.z), lo(a.w)), pk2(lo(b.x), lo(b.y)), pk2(lo(b.z), lo(b.w)));
            }
        }
        __syncthreads();
        if (CL == 1 && hpart) ring_prime(ring, KSpp, w.g_h, w.g_l, NTc, wave, lane, 0, KSp, hpart * KSpp);
        ring_gemm(acc, ring, s_ph, s_pl, sp, KSpp, w.g_h, w.g_l, lane, CL == 1 && hpart != 0);
        __syncthreads();                               // every wave is done reading this part of pooled
    }
    LT_STAMP(2);
    ring_prime(ring, KSx, w.wk_h, w.wk_l, NTh, wave, lane, 0);           // w_k flies through the conversion / exchange of x
    if (CL > 1) {
        // partial [tile][rank][wave][t][lane] x 16 B: a lane writes and reads exactly the accumulator slots it owns
        const __amdgpu_buffer_rsrc_t xp = __builtin_amdgcn_make_buffer_rsrc(xpart + (size_t)tile * CL * (8 * 3 * 64 * 4), 0,
                                                                            CL * 8 * 3 * 64 * 16, 0x00027000);
        const int slot = (wave * 3 * 64 + lane) * 16;
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (wave + 8 * t < NTc)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lt_i32x4, acc[t]), xp, slot + t * 64 * 16, crank * (8 * 3 * 64 * 16), 17);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's write-through stores are acknowledged
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(&counters[2 * tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // The ranks of a tile are ADJACENT in dispatch order, so at most one cluster of a launch straddles the edge of what is
            // resident and everything in front of it retires without waiting for anybody: no co-residency requirement beyond
            // in-order dispatch.  The wait is bounded all the same (a device in trouble must not become a hang): when it runs out
            // the rank goes on with what has arrived and raises the library's status word (mgnns_set_status_word).
            int spins = 0;
            while (__hip_atomic_load(&counters[2 * tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < CL) {
                if (++spins > (1 << 24)) {
                    if (status) __hip_atomic_store(status, MGNNS_STATUS_CLUSTER_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            f32x4 pr[CL];
#pragma unroll
            for (int rk = 0; rk < CL; ++rk)
                pr[rk] = wave + 8 * t < NTc ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xp, slot + t * 64 * 16, rk * (8 * 3 * 64 * 16), 17))
                                            : f32x4{0.f, 0.f, 0.f, 0.f};
            acc[t] = pr[0];
#pragma unroll
            for (int rk = 1; rk < CL; ++rk) acc[t] += pr[rk];
        }
        // (the loads above are complete once acc is consumed below; the departure count follows the conversion barrier)
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int nt = wave + 8 * t, n = nt * 16 + ccol;
        if (nt < NTc && n < C) {
#pragma unroll
            for (int r = 0; r < 4; ++r) put(xh16, xl16, (crow + r) * sxc * 8 + n, acc[t][r]);
        }
    }
    for (int i = tid; i < (1 + LO) * LT_ROWS * spf; i += LT_THR) s_ph[i] = make_uint4(0u, 0u, 0u, 0u);   // -> flatten buffer, zero padded
    __syncthreads();
    if (CL > 1 && tid == 0) {      // every thread of this rank has consumed the partials: the last rank to get here re-arms the counters
        const int old = __hip_atomic_fetch_add(&counters[2 * tile + 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == CL - 1) {
            __hip_atomic_store(&counters[2 * tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&counters[2 * tile + 1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- K, V ----------------------------------------------------------------------------------------------------------------
    LT_STAMP(3);
    ring_gemm(acc, ring, s_xh, s_xl, sxc, KSx, w.wk_h, w.wk_l, lane);
    ring_prime(ring, KSx, w.wv_h, w.wv_l, NTh, wave, lane, 0);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int nt = wave + 8 * t, n = nt * 16 + ccol;
        if (nt < NTh && n < hid) {
            const float b0 = w.bk[n];
#pragma unroll
            for (int r = 0; r < 4; ++r) s_k[(crow + r) * shf + n] = acc[t][r] + b0;
        }
    }
    ring_gemm(acc, ring, s_xh, s_xl, sxc, KSx, w.wv_h, w.wv_l, lane);
    ring_prime(ring, KSf, w.xl_h, w.xl_l, NTo, wave, lane, 0);            // x_linear's first k-steps fly through the label loop
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int nt = wave + 8 * t, n = nt * 16 + ccol;
        if (nt < NTh && n < hid) {
            const float b0 = w.bv[n];
#pragma unroll
            for (int r = 0; r < 4; ++r) s_v[(crow + r) * shf + n] = acc[t][r] + b0;
        }
    }
    // the composed map's fragments of this wave's column tile stay in registers across the NLQ label rows
    uint4 wch[LB_MAXKS], wcl[LO ? LB_MAXKS : 1];
    if (wave < NT5) {
#pragma unroll
        for (int ks = 0; ks < LB_MAXKS; ++ks) {
            const size_t off = ((size_t)wave * KSh + (ks < KSh ? ks : 0)) * 64 + lane;
            wch[ks] = reinterpret_cast<const uint4*>(w.wc_h)[off];
            if (LO) wcl[ks] = reinterpret_cast<const uint4*>(w.wc_l)[off];
        }
    }
    __syncthreads();

    // ---- per label row l: element-wise attention (fp32) into s_o, composed map into its slot of the flatten buffer -----------
    const float inv_scale = 1.0f / sqrtf((float)dh);
    LT_STAMP(4);
    for (int l = 0; l < NLQ; ++l) {
        // wave w owns sample rows 2w, 2w+1; a lane holds dim `lane` of EVERY head (independent reductions interleave:
        // one head at a time was a ~400-cycle dependent chain per (row, head), 70 of them per wave)
#pragma unroll 1
        for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * wave + rr;
            const bool on = lane < dh;
            float e[LT_MAXH], vv[LT_MAXH];
#pragma unroll
            for (int h = 0; h < LT_MAXH; ++h) {
                const int c = (h < n_heads ? h : 0) * dh + (on ? lane : 0);
                e[h] = (on && h < n_heads) ? s_q[l * hid + c] * s_k[r * shf + c] * inv_scale : -INFINITY;
                vv[h] = s_v[r * shf + c];
            }
#pragma unroll
            for (int h = 0; h < LT_MAXH; ++h) {
                if (h < n_heads) {                                  // wave-uniform
                    const float m = wave_max_dpp(e[h]);
                    const float p = on ? __expf(e[h] - m) : 0.f;
                    const float z = wave_sum_dpp(p);
                    if (on) put(oh16, ol16, r * shc * 8 + h * dh + lane, (p / z) * vv[h]);
                }
            }
        }
        __syncthreads();
        if (wave < NT5) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            const int aoff = (lane & 15) * shc + (lane >> 4);
#pragma unroll
            for (int ks = 0; ks < LB_MAXKS; ++ks)
                if (ks < KSh) {
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, s_oh[aoff + ks * 4]);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, wch[ks]), a, 0, 0, 0);
                    if (LO) {
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, wcl[ks]), a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, s_ol[aoff + ks * 4]),
                                                                    __builtin_bit_cast(bf16x8, wch[ks]), a, 0, 0, 0);
                    }
                }
            const int n = wave * 16 + ccol;
            if (n < N5) {
                const float b0 = w.bc[n];
#pragma unroll
                for (int r = 0; r < 4; ++r) put(fh16, fl16, (crow + r) * sfc * 8 + l * N5 + n, a[r] + b0);
            }
        }
        __syncthreads();
    }

    // ---- out = x_linear(flat) -------------------------------------------------------------------------------------------------
    LT_STAMP(5);
    ring_gemm(acc, ring, s_fh, s_fl, sfc, KSf, w.xl_h, w.xl_l, lane);
    LT_STAMP(6);
    // the next query's column-tile slots (8 tiles each) are divided over the cluster ranks
    const int q_slots = ((HKn + 15) / 16 + 7) / 8, q_per = (q_slots + CL - 1) / CL;
    const int q_lo = crank * q_per, q_hi = q_lo + q_per < q_slots ? q_lo + q_per : q_slots;
    if (w.wq_h) ring_prime(ring, KSh, w.wq_h, w.wq_l, (HKn + 15) / 16, wave, lane, q_lo);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int nt = wave + 8 * t, n = nt * 16 + ccol;
        if (nt < NTo && n < NO) {
            const float b0 = w.bxl[n];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = r0 + crow + r;
                const float v = acc[t][r] + b0;
                if (gr < B && crank == 0) out[(size_t)gr * NO + n] = v;
                if (w.wq_h) put(oh16, ol16, (crow + r) * shc * 8 + n, v);          // NO == hid (launcher)
            }
        }
    }
    // ---- qh = w_qs(out) + b -----------------------------------------------------------------------------------------------------
    if (w.wq_h) {
        __syncthreads();
        const int NTq = (HKn + 15) / 16;
        for (int t0 = q_lo; t0 < q_hi; t0 += 3) {
            if (t0 != q_lo) ring_prime(ring, KSh, w.wq_h, w.wq_l, NTq, wave, lane, t0);
            ring_gemm(acc, ring, s_oh, s_ol, shc, KSh, w.wq_h, w.wq_l, lane);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int nt = wave + 8 * (t0 + t), n = nt * 16 + ccol;
                if (t0 + t < q_hi && nt < NTq && n < HKn) {
                    const float b0 = w.bq ? w.bq[n] : 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int gr = r0 + crow + r;
                        if (gr < B) qh_next[(size_t)gr * HKn + n] = acc[t][r] + b0;
                    }
                }
            }
        }
    }
#if MG_LT_TOUCH
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(touch_sink)::"memory");      // (long landed: keeps the register reserved up to here)
#endif
    LT_STAMP(7);
}

}  // namespace

// The shape limits of mgnns_label_tail_bf16_fwd as a predicate (see mgnns_label_tail_supported).
extern "C" int mgnns_label_tail_bf16_supported(int C, int NLQ, int n_heads, int dh, int N5, int n_out, int K_pool, int terms,
                                               int with_next_q) {
    if (!(C > 0 && C <= 384 && NLQ > 0 && n_heads > 0 && dh > 0 && dh <= 64 && N5 > 0 && N5 <= 128 && n_out > 0)) return 0;
    if (terms != 1 && terms != 3) return 0;
    if (!(K_pool > 0 && K_pool % 128 == 0)) return 0;
    const int hid = n_heads * dh;
    if (!(hid <= 32 * LB_MAXKS && n_out <= 384 && n_heads <= LT_MAXH)) return 0;
    if (with_next_q && n_out != hid) return 0;
    const int nh = terms == 3 ? 4 : 1, lo = terms == 3 ? 1 : 0;
    const int sp = 4 * (K_pool / 32 / nh) + 2, sxc = 4 * ((C + 31) / 32) + 2, shc = 4 * ((hid + 31) / 32) + 2, sfc = 4 * ((NLQ * N5 + 31) / 32) + 2;
    const size_t lds = (size_t)(1 + lo) * LT_ROWS * ((sp > sfc ? sp : sfc) + sxc + shc) * 16 +
                       ((size_t)2 * LT_ROWS * lt_stride(hid) + (size_t)NLQ * hid) * sizeof(float);
    return lds <= 160 * 1024 ? 1 : 0;
}

extern "C" int mgnns_label_tail_bf16_fwd(const float* pooled, int B, int n_parts, int K_pool, int C, int terms,
                                         const void* const* packed /* g, wk, wv, wc, xl, wq_next: (hi, lo) pairs */,
                                         const float* Q, int NLQ, int n_heads, int dh, const float* bk, const float* bv,
                                         const float* bc, int N5, const float* bxl, int n_out, float* out,
                                         const float* bq_next, int HK_next, float* qh_next, float* cluster_scratch,
                                         int* cluster_counters, mgnns_stream_t stream) {
    MG_REQUIRE(B >= 0 && C > 0 && C <= 384 && NLQ > 0 && n_heads > 0 && dh > 0 && dh <= 64 && N5 > 0 && N5 <= 128 && n_out > 0,
               "mgnns_label_tail_bf16_fwd: bad dims B=%d C=%d (<= 384) NLQ=%d heads=%d dh=%d N5=%d out=%d", B, C, NLQ, n_heads, dh, N5, n_out);
    MG_REQUIRE(terms == 1 || terms == 3, "mgnns_label_tail_bf16_fwd: terms must be 1 (bf16) or 3 (split-bf16)");
    if (B == 0) return 0;
    MG_REQUIRE(pooled && packed && Q && bk && bv && bc && bxl && out, "mgnns_label_tail_bf16_fwd: null pointer");
    for (int i = 0; i < 10; ++i) MG_REQUIRE(packed[i], "mgnns_label_tail_bf16_fwd: packed weight %d missing", i);
    MG_REQUIRE(n_parts >= 1 && K_pool > 0 && K_pool % 128 == 0 && mg_aligned16(pooled),
               "mgnns_label_tail_bf16_fwd: pooled [B,%d,%d] must be 16-byte aligned with K %% 128 == 0", n_parts, K_pool);
    const int hid = n_heads * dh;
    MG_REQUIRE(hid <= 32 * LB_MAXKS && n_out <= 384 && n_heads <= LT_MAXH,
               "mgnns_label_tail_bf16_fwd: hidden width %d / output width %d / %d heads unsupported", hid, n_out, n_heads);
    MG_REQUIRE(!packed[10] || (packed[11] && qh_next && HK_next > 0 && n_out == hid),
               "mgnns_label_tail_bf16_fwd: the query projection needs both packed buffers, qh_next, HK_next and n_out == hidden width");
    LabelW w;
    const unsigned short* const* pk = reinterpret_cast<const unsigned short* const*>(packed);
    w.g_h = pk[0]; w.g_l = pk[1]; w.wk_h = pk[2]; w.wk_l = pk[3]; w.wv_h = pk[4]; w.wv_l = pk[5]; w.wc_h = pk[6]; w.wc_l = pk[7];
    w.xl_h = pk[8]; w.xl_l = pk[9]; w.wq_h = pk[10]; w.wq_l = pk[11];
    w.bk = bk; w.bv = bv; w.bc = bc; w.bxl = bxl; w.bq = bq_next;
    const int nh = terms == 3 ? 4 : 1, lo = terms == 3 ? 1 : 0;
    const int sp = 4 * (K_pool / 32 / nh) + 2, sxc = 4 * ((C + 31) / 32) + 2, shc = 4 * ((hid + 31) / 32) + 2, sfc = 4 * ((NLQ * N5 + 31) / 32) + 2;
    const size_t lds = (size_t)(1 + lo) * LT_ROWS * ((sp > sfc ? sp : sfc) + sxc + shc) * 16 +
                       ((size_t)2 * LT_ROWS * lt_stride(hid) + (size_t)NLQ * hid) * sizeof(float);
    MG_REQUIRE(lds <= 160 * 1024, "mgnns_label_tail_bf16_fwd: K_pool=%d, C=%d need %zu B of LDS (> 160 KiB)", K_pool, C, lds);
    MG_REQUIRE((cluster_scratch != nullptr) == (cluster_counters != nullptr), "mgnns_label_tail_bf16_fwd: cluster scratch and counters go together");
    MG_REQUIRE(!cluster_scratch || mg_aligned16(cluster_scratch), "mgnns_label_tail_bf16_fwd: cluster scratch must be 16-byte aligned");
    if (cluster_scratch)
        if (int rc = mg_check_status("mgnns_label_tail_bf16_fwd")) return rc;   // a bounded wait of an earlier persistent launch ran out
    MG_DYN_LDS((label_tail_bf16_kernel<1, 1>), 160 * 1024);
    MG_DYN_LDS((label_tail_bf16_kernel<3, 1>), 160 * 1024);
    MG_DYN_LDS((label_tail_bf16_kernel<3, 4>), 160 * 1024);
    const int tiles = (B + LT_ROWS - 1) / LT_ROWS;
    const dim3 blk(LT_THR);
    if (terms == 3 && cluster_scratch)
        hipLaunchKernelGGL((label_tail_bf16_kernel<3, 4>), dim3(tiles * 4), blk, lds, (hipStream_t)stream, pooled, B, n_parts, K_pool, C, Q, NLQ,
                           n_heads, dh, w, N5, n_out, out, HK_next, qh_next, cluster_scratch, cluster_counters, mg_status_word());
    else if (terms == 3)
        hipLaunchKernelGGL((label_tail_bf16_kernel<3, 1>), dim3(tiles), blk, lds, (hipStream_t)stream, pooled, B, n_parts, K_pool, C, Q, NLQ, n_heads, dh,
                           w, N5, n_out, out, HK_next, qh_next, (float*)nullptr, (int*)nullptr, (int*)nullptr);
    else
        hipLaunchKernelGGL((label_tail_bf16_kernel<1, 1>), dim3(tiles), blk, lds, (hipStream_t)stream, pooled, B, n_parts, K_pool, C, Q, NLQ, n_heads, dh,
                           w, N5, n_out, out, HK_next, qh_next, (float*)nullptr, (int*)nullptr, (int*)nullptr);
    MG_CHECK_LAUNCH("mgnns_label_tail_bf16_fwd");
    return 0;
}
