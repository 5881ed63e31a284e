// Single-query multi-head attention with the K/V projections fused in (submodules.py:55-119,
// len_q == 1): one workgroup per (sample, head).  K_h = bank W_k,h^T and V_h = bank W_v,h^T are
// produced tile by tile on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32) and consumed from the
// accumulators (score dot / softmax-weighted sum); they never touch HBM.
//
// Geometry: 4 waves, wave w owns head dims [32w, 32w+32) (2 column tiles) for up to 13 row tiles
// (L <= 208).  Row tiles entirely behind the last unmasked position are skipped: a masked position
// gets probability exactly 0 (softmax of -inf), so its K/V rows cannot influence the output.
#include "common.hpp"

namespace {

constexpr int MT = 13;                   // row tiles (L <= 208)
constexpr int LMAX = MT * 16;
constexpr int BK = 20;                   // K-slice of the model dim (300 = 15 x 20)
constexpr int SX = BK + 2;               // LDS row stride 22: rows 0..15 x k{0,1} hit 32 distinct banks
constexpr int DK = 128;

// one BK-slice of MFMAs for a COMPILE-TIME number of row tiles: all A fragments of a k-step are requested before
// the MFMAs that consume them (no per-tile branch, no LDS round trip in front of every MFMA pair)
template <int NMT>
__device__ __forceinline__ void slice_mma(f32x4 (&acc)[MT][2], const float* __restrict__ xs, const float* __restrict__ ws,
                                          int wave, int lane) {
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
        const float* ap = xs + (lane & 15) * SX + kk + (lane >> 4);
        const float* bp = ws + (wave * 32 + (lane & 15)) * SX + kk + (lane >> 4);
        const float b0 = bp[0], b1 = bp[16 * SX];
        float a[NMT];
#pragma unroll
        for (int i = 0; i < NMT; ++i) a[i] = ap[i * 16 * SX];
#pragma unroll
        for (int i = 0; i < NMT; ++i) {
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b0, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b1, acc[i][1], 0, 0, 0);
        }
    }
}

__global__ __launch_bounds__(256) void sq_mha_core_kernel(const float* __restrict__ qh, const float* __restrict__ bank,
                                                          const float* __restrict__ mask, int B, int L, int D, int H,
                                                          const float* __restrict__ Wk, const float* __restrict__ bk,
                                                          const float* __restrict__ Wv, const float* __restrict__ bv,
                                                          float inv_temp_div, float* __restrict__ o,
                                                          float* __restrict__ attn) {
    __shared__ __attribute__((aligned(16))) float Xs[2][LMAX * SX];
    __shared__ __attribute__((aligned(16))) float Ws[2][DK * SX];
    __shared__ float s_part[4][LMAX];
    __shared__ float s_p[LMAX];
    __shared__ float s_red[8];
    __shared__ int s_lvalid;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const float* xb = bank + (size_t)b * L * D;
    const float* qv = qh + (size_t)b * H * DK + h * DK;

    // ---- last unmasked position -> number of live row tiles ------------------------------------
    if (tid == 0) s_lvalid = mask ? 0 : L;
    __syncthreads();
    if (mask) {
        int last = 0;
        for (int t = tid; t < L; t += 256)
            if (mask[(size_t)b * L + t] != 0.0f) last = t + 1;
        if (last) atomicMax(&s_lvalid, last);
        __syncthreads();
    }
    const int lvalid = s_lvalid;
    const int n_mt = (lvalid + 15) >> 4;
    // tile-count class of the branch-free MFMA body; rows between n_mt*16 and n_sel*16 hold real (masked) or zero
    // data and end up with probability 0
    const int n_sel = n_mt <= 1 ? 1 : n_mt <= 2 ? 2 : n_mt <= 4 ? 4 : n_mt <= 7 ? 7 : n_mt <= 10 ? 10 : MT;
    const int rows_live = n_sel * 16;
    const int nchunk = (D + BK - 1) / BK;

    f32x4 acc[MT][2];
    float r_o[2] = {0.f, 0.f};

    for (int phase = 0; phase < 2; ++phase) {
        const float* Wm = (phase == 0 ? Wk : Wv) + (size_t)h * DK * D;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 rx[5], rw[3];
        auto gload = [&](int k0) {
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int q = tid + u * 256;             // float4 id: row = q / 5, c4 = q % 5
                const int row = q / 5, c4 = q - row * 5;
                const int k = k0 + 4 * c4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row < rows_live && row < L && k < D) v = *reinterpret_cast<const f32x4*>(xb + (size_t)row * D + k);
                rx[u] = v;
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int q = tid + u * 256;
                const int row = q / 5, c4 = q - row * 5;
                const int k = k0 + 4 * c4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row < DK && k < D) v = *reinterpret_cast<const f32x4*>(Wm + (size_t)row * D + k);
                rw[u] = v;
            }
        };
        auto lstore = [&](int buf) {
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int q = tid + u * 256;
                const int row = q / 5, c4 = q - row * 5;
                if (row < rows_live) {
                    float* d = &Xs[buf][row * SX + 4 * c4];
                    *reinterpret_cast<float2*>(d) = float2{rx[u][0], rx[u][1]};
                    *reinterpret_cast<float2*>(d + 2) = float2{rx[u][2], rx[u][3]};
                }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int q = tid + u * 256;
                const int row = q / 5, c4 = q - row * 5;
                if (row < DK) {
                    float* d = &Ws[buf][row * SX + 4 * c4];
                    *reinterpret_cast<float2*>(d) = float2{rw[u][0], rw[u][1]};
                    *reinterpret_cast<float2*>(d + 2) = float2{rw[u][2], rw[u][3]};
                }
            }
        };

        gload(0);
        lstore(0);
        __syncthreads();
        for (int c = 0; c < nchunk; ++c) {
            const int buf = c & 1;
            if (c + 1 < nchunk) gload((c + 1) * BK);
            switch (n_sel) {
                case 1: slice_mma<1>(acc, Xs[buf], Ws[buf], wave, lane); break;
                case 2: slice_mma<2>(acc, Xs[buf], Ws[buf], wave, lane); break;
                case 4: slice_mma<4>(acc, Xs[buf], Ws[buf], wave, lane); break;
                case 7: slice_mma<7>(acc, Xs[buf], Ws[buf], wave, lane); break;
                case 10: slice_mma<10>(acc, Xs[buf], Ws[buf], wave, lane); break;
                default: slice_mma<MT>(acc, Xs[buf], Ws[buf], wave, lane); break;
            }
            if (c + 1 < nchunk) lstore(buf ^ 1);
            __syncthreads();
        }

        if (phase == 0) {
            // ---- scores: s[l] = (q_h . (K[l,:] + bk_h)) / temperature ------------------------------
            const int c0 = wave * 32 + (lane & 15);
            const float q0 = qv[c0], q1 = qv[c0 + 16];
            const float k0b = bk ? bk[h * DK + c0] : 0.f, k1b = bk ? bk[h * DK + c0 + 16] : 0.f;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if (i < n_mt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = q0 * (acc[i][0][r] + k0b) + q1 * (acc[i][1][r] + k1b);
                        v += __shfl_xor(v, 1, 64);
                        v += __shfl_xor(v, 2, 64);
                        v += __shfl_xor(v, 4, 64);
                        v += __shfl_xor(v, 8, 64);
                        if ((lane & 15) == 0) s_part[wave][i * 16 + (lane >> 4) * 4 + r] = v;
                    }
                }
            }
            __syncthreads();
            // ---- masked softmax over l (one position per thread, L <= 208 < 256) --------------------
            float s = -INFINITY;
            if (tid < lvalid) {
                s = (s_part[0][tid] + s_part[1][tid] + s_part[2][tid] + s_part[3][tid]) / inv_temp_div;
                if (mask && mask[(size_t)b * L + tid] == 0.0f) s = -INFINITY;
            }
            float m = wave_max(s);
            if (lane == 0) s_red[wave] = m;
            __syncthreads();
            m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
            const float e = (tid < lvalid && s != -INFINITY) ? expf(s - m) : 0.f;
            float z = wave_sum(e);
            if (lane == 0) s_red[4 + wave] = z;
            __syncthreads();
            z = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
            const float p = e / z;
            if (tid < LMAX) s_p[tid] = p;
            if (attn && tid < L) attn[((size_t)h * B + b) * L + tid] = p;
            __syncthreads();
        } else {
            // ---- o[c] = sum_l p[l] * (V[l,c] + bv[c]) ---------------------------------------------------
            const int c0 = wave * 32 + (lane & 15);
            const float v0b = bv ? bv[h * DK + c0] : 0.f, v1b = bv ? bv[h * DK + c0 + 16] : 0.f;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if (i < n_mt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = s_p[i * 16 + (lane >> 4) * 4 + r];
                        r_o[0] = fmaf(p, acc[i][0][r] + v0b, r_o[0]);
                        r_o[1] = fmaf(p, acc[i][1][r] + v1b, r_o[1]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                r_o[j] += __shfl_xor(r_o[j], 16, 64);
                r_o[j] += __shfl_xor(r_o[j], 32, 64);
            }
            if (lane < 16) {
                float* ob = o + (size_t)b * H * DK + h * DK + wave * 32 + lane;
                ob[0] = r_o[0];
                ob[16] = r_o[1];
            }
        }
    }
}

}  // namespace

extern "C" int mgnns_sq_mha_core_fwd(const float* qh, const float* bank, const float* mask, int B, int L, int D,
                                     int H, int dk, const float* Wk, const float* bk, const float* Wv,
                                     const float* bv, float* o, float* attn, mgnns_stream_t stream) {
    MG_REQUIRE(qh && bank && Wk && Wv && o, "mgnns_sq_mha_core_fwd: null pointer");
    MG_REQUIRE(dk == DK, "mgnns_sq_mha_core_fwd: d_kv=%d unsupported (128 only)", dk);
    MG_REQUIRE(B >= 0 && H > 0 && L > 0 && L <= LMAX, "mgnns_sq_mha_core_fwd: L=%d unsupported (1..%d)", L, LMAX);
    MG_REQUIRE(D > 0 && D % 4 == 0 && D <= 320, "mgnns_sq_mha_core_fwd: D=%d unsupported (multiple of 4, <= 320)", D);
    MG_REQUIRE(mg_aligned16(bank) && mg_aligned16(Wk) && mg_aligned16(Wv),
               "mgnns_sq_mha_core_fwd: bank/Wk/Wv must be 16-byte aligned");
    if (B == 0) return 0;
    const float temp = (float)sqrt((double)dk);      // np.power(d_k, 0.5), submodules.py:31
    hipLaunchKernelGGL(sq_mha_core_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, qh, bank, mask, B, L, D, H,
                       Wk, bk, Wv, bv, temp, o, attn);
    MG_CHECK_LAUNCH("mgnns_sq_mha_core_fwd");
    return 0;
}
