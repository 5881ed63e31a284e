// Generic fp32 GEMM on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32): the small dense layers of the path
// (nn.Linear / Conv1d(k=1) / GraphConvolution X*W / read-out / LSTM input projection).
//   NT: Y = act(X[M,K] * W[N,K]^T + bias) + residual      (nn.Linear weight layout)
//   NN: Y = act(X[M,K] * W[K,N])                          (GraphConvolution weight layout)
// 64x64 block tile, 4 waves as 2x2, each wave 2x2 MFMA tiles of 16x16, BK = 32, LDS-staged operands with a
// register prefetch of the next K-slice (one barrier pair per slice, global latency under the MFMAs).
// These layers are tiny (M = batch = 256): a 64x64 tiling alone yields 8-80 workgroups on a 256-CU chip and a
// serial K loop, so the K dimension is split across gridDim.z workgroups (partials in a workspace) and a second
// small kernel reduces the slices in a fixed order and applies bias / activation / residual.
#include "common.hpp"

namespace {

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int SA = BK + 2;    // [row][k] stride 34 floats: rows 0..15 x k{0,1} -> 32 distinct banks (ds_read_b32)
constexpr int SBN = BN + 16;  // [k][n] stride for the NN form: 80 floats

__device__ __forceinline__ f32x4 load4(const float* p, int n_valid, bool vec_ok) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n_valid >= 4 && vec_ok) {
        v = *reinterpret_cast<const f32x4*>(p);
    } else {
        if (n_valid > 0) v[0] = p[0];
        if (n_valid > 1) v[1] = p[1];
        if (n_valid > 2) v[2] = p[2];
        if (n_valid > 3) v[3] = p[3];
    }
    return v;
}

// strided batch descriptor (element strides between consecutive problems); n == 0: plain / K-split launch
struct GemmBatch { int n; long sx, sw, sy, sb; };

// grid (ceil(N/64), ceil(M/64), S).  S == 1: full epilogue here.  S > 1: raw partial sums of K-slice z go to
// part[z][M][N] and gemm_reduce_kernel finishes.
template <bool W_IS_KN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ X, int M, int K,
                                                       const float* __restrict__ W, int N,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ residual,
                                                       float* __restrict__ Y, int ldy, int act, int vecX, int vecW,
                                                       const int32_t* __restrict__ gather_idx,
                                                       const int32_t* __restrict__ m_dev, int kslice,
                                                       float* __restrict__ part, int ldx, GemmBatch bt) {
    __shared__ __attribute__((aligned(16))) float As[BM * SA];
    __shared__ __attribute__((aligned(16))) float Bs[W_IS_KN ? BK * SBN : BN * SA];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if (m_dev) {                      // device-side row count (packed ragged rows): M is only the upper bound
        M = min(M, *m_dev);
        if (m0 >= M) return;
    }
    if (bt.n) {                       // strided batch: blockIdx.z selects the problem, no K split
        X += (size_t)blockIdx.z * bt.sx;
        W += (size_t)blockIdx.z * bt.sw;
        Y += (size_t)blockIdx.z * bt.sy;
        if (bias) bias += (size_t)blockIdx.z * bt.sb;
    }
    const int kbeg = bt.n ? 0 : blockIdx.z * kslice;
    const int kend = min(K, kbeg + kslice);

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging map (A, and B in NT form): 64 rows x 8 float4 per slice = 512 float4, two per thread
    const int s_row = tid >> 3, s_k4 = (tid & 7) * 4;                    // rows s_row and s_row + 32
    size_t a_row[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int gm = m0 + s_row + 32 * u;
        a_row[u] = gm < M ? (gather_idx ? (size_t)gather_idx[gm] : (size_t)gm) : 0;
    }
    // NN form: 32 k-rows x 16 float4 = 512 float4, two per thread
    const int b_kr = tid >> 4, b_n4 = (tid & 15) * 4;                    // k-rows b_kr and b_kr + 16

    f32x4 ra[2], rb[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int gm = m0 + s_row + 32 * u, gk = k0 + s_k4;
            ra[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (gm < M) ra[u] = load4(X + a_row[u] * ldx + gk, kend - gk, vecX);
            if (!W_IS_KN) {
                const int gn = n0 + s_row + 32 * u;
                rb[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (gn < N) rb[u] = load4(W + (size_t)gn * K + gk, kend - gk, vecW);
            } else {
                const int gkk = k0 + b_kr + 16 * u, gn = n0 + b_n4;
                rb[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (gkk < kend) rb[u] = load4(W + (size_t)gkk * N + gn, N - gn, vecW);
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float* d = &As[(s_row + 32 * u) * SA + s_k4];
            *reinterpret_cast<float2*>(d) = float2{ra[u][0], ra[u][1]};
            *reinterpret_cast<float2*>(d + 2) = float2{ra[u][2], ra[u][3]};
            if (!W_IS_KN) {
                float* e = &Bs[(s_row + 32 * u) * SA + s_k4];
                *reinterpret_cast<float2*>(e) = float2{rb[u][0], rb[u][1]};
                *reinterpret_cast<float2*>(e + 2) = float2{rb[u][2], rb[u][3]};
            } else {
                *reinterpret_cast<f32x4*>(&Bs[(b_kr + 16 * u) * SBN + b_n4]) = rb[u];
            }
        }
    };

    gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        lstore();
        __syncthreads();
        if (k0 + BK < kend) gload(k0 + BK);          // next slice in flight under this slice's MFMAs
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[i] = As[(wr * 32 + i * 16 + (lane & 15)) * SA + kk + (lane >> 4)];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = wc * 32 + j * 16 + (lane & 15);
                b[j] = W_IS_KN ? Bs[(kk + (lane >> 4)) * SBN + n] : Bs[n * SA + kk + (lane >> 4)];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: C layout col = lane&15, row = (lane>>4)*4 + r
    const bool partial = !bt.n && gridDim.z > 1;
    float* pz = partial ? part + (size_t)blockIdx.z * M * N : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gn = n0 + wc * 32 + j * 16 + (lane & 15);
            if (gn >= N) continue;
            const float bv = (!partial && bias) ? bias[gn] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wr * 32 + i * 16 + (lane >> 4) * 4 + r;
                if (gm >= M) continue;
                if (partial) {
                    pz[(size_t)gm * N + gn] = acc[i][j][r];
                } else {
                    float v = mg_act(acc[i][j][r] + bv, act);
                    if (residual) v += residual[(size_t)gm * N + gn];
                    Y[(size_t)gm * ldy + gn] = v;
                }
            }
        }
}

// Small problems (a few hundred rows, K of a few hundred): latency, not throughput, decides.  One wave per 16x16
// output tile, no LDS and no barrier: the wave fetches its A rows and B columns straight into MFMA operand registers
// -- with K walked in the permuted order (16j + 4g + e) every lane's fetch is 16 contiguous bytes -- so all loads of
// a 80-deep K chunk are in flight together and the MFMAs follow back to back.  Operand tiles are re-read by
// neighbouring waves from L2, which is irrelevant at these sizes.  Requires K % 4 == 0 and 16-byte aligned rows.
// grid (ceil(N/64), ceil(M/16), nbatch or 1), 4 waves = 4 adjacent column tiles.
template <bool W_IS_KN>
__global__ __launch_bounds__(256) void gemm_small_kernel(const float* __restrict__ X, int M, int K,
                                                         const float* __restrict__ W, int N,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ residual, float* __restrict__ Y,
                                                         int ldy, int act, int ldx, GemmBatch bt) {
    if (bt.n) {
        X += (size_t)blockIdx.z * bt.sx;
        W += (size_t)blockIdx.z * bt.sw;
        Y += (size_t)blockIdx.z * bt.sy;
        if (bias) bias += (size_t)blockIdx.z * bt.sb;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * 16, n0 = (blockIdx.x * 4 + wave) * 16;
    if (n0 >= N) return;
    const int row = m0 + n, col = n0 + n;
    const bool rv = row < M, cv = col < N;
    const float* xr = X + (size_t)(rv ? row : 0) * ldx;
    const float* wr = W_IS_KN ? W + (cv ? col : 0) : W + (size_t)(cv ? col : 0) * K;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    constexpr int CH = 5;                             // k-groups of 16 per chunk
    for (int k0 = 0; k0 < K; k0 += 16 * CH) {
        f32x4 a[CH], w[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int kk = k0 + 16 * j + 4 * g;
            a[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            w[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kk < K) {                             // K % 4 == 0: the whole group of four is inside
                if (rv) a[j] = *reinterpret_cast<const f32x4*>(xr + kk);
                if (cv) {
                    if (W_IS_KN) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[j][e] = wr[(size_t)(kk + e) * N];
                    } else {
                        w[j] = *reinterpret_cast<const f32x4*>(wr + kk);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc[e & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][e], w[j][e], acc[e & 1], 0, 0, 0);
    }
    // C layout: col = lane & 15, rows 4g + r
    if (!cv) return;
    const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gm = m0 + 4 * g + r;
        if (gm >= M) continue;
        float v = mg_act(acc[0][r] + acc[1][r] + bv, act);
        if (residual) v += residual[(size_t)gm * N + col];
        Y[(size_t)gm * ldy + col] = v;
    }
}

// the small-problem kernel pays when the tiled kernel could not fill the chip anyway
static bool use_small(int M, int N, int K, int ldx, const float* X, const float* W, bool w_is_kn, long sx, long sw) {
    if (K % 4 || ldx % 4 || sx % 4 || !mg_aligned16(X) || K > 512) return false;
    if (!w_is_kn && (sw % 4 || !mg_aligned16(W))) return false;
    return (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN) <= 96;
}

// Y = act(sum_z part[z] + bias) + residual, slices summed in ascending z (deterministic)
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const float* __restrict__ part, int S, int M, int N,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ residual, float* __restrict__ Y,
                                                          int ldy, int act) {
    const size_t total = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t m = i / N;
        const int n = (int)(i - m * N);
        float s = part[i];
        for (int z = 1; z < S; ++z) s += part[(size_t)z * total + i];
        float v = mg_act(s + (bias ? bias[n] : 0.f), act);
        if (residual) v += residual[i];
        Y[m * ldy + n] = v;
    }
}

// K-split factor: enough workgroups to cover the chip, slices of at least 2 x BK, bounded by the workspace
int choose_split(int M, int N, int K, size_t ws_floats) {
    const int tiles = ((N + BN - 1) / BN) * ((M + BM - 1) / BM);
    // short reductions (K < 512: <= 16 slices of BK) finish in a few us on their own; a K split would only add the
    // reduce launch (~4.5 us inside a graph)
    if (tiles >= 192 || K < 512) return 1;
    int s = (384 + tiles - 1) / tiles;
    const int smax = K / (2 * BK);
    if (s > smax) s = smax;
    if (s > 16) s = 16;
    while (s > 1 && (size_t)s * M * N > ws_floats) --s;
    return s < 1 ? 1 : s;
}

template <bool W_IS_KN>
int launch_gemm(const float* X, int M, int K, const float* W, int N, const float* bias, const float* residual, float* Y,
                int ldy, int act, const int32_t* gather_idx, const int32_t* m_dev, float* ws, size_t ws_floats,
                hipStream_t stream) {
    if (!gather_idx && !m_dev && use_small(M, N, K, K, X, W, W_IS_KN, 0, 0)) {
        hipLaunchKernelGGL(gemm_small_kernel<W_IS_KN>, dim3((N + 63) / 64, (M + 15) / 16, 1), dim3(256), 0, stream, X, M, K, W,
                           N, bias, residual, Y, ldy, act, K, GemmBatch{0, 0, 0, 0, 0});
        return 0;
    }
    const int vecX = (K % 4 == 0) && mg_aligned16(X);
    const int vecW = W_IS_KN ? ((N % 4 == 0) && mg_aligned16(W)) : ((K % 4 == 0) && mg_aligned16(W));
    int S = (ws && !m_dev) ? choose_split(M, N, K, ws_floats) : 1;
    int kslice = K;
    if (S > 1) {
        kslice = (((K + S - 1) / S) + BK - 1) / BK * BK;
        S = (K + kslice - 1) / kslice;
    }
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, S);
    hipLaunchKernelGGL(gemm_f32_kernel<W_IS_KN>, grid, dim3(256), 0, stream, X, M, K, W, N, bias, residual, Y, ldy, act,
                       vecX, vecW, gather_idx, m_dev, kslice, ws, K, GemmBatch{0, 0, 0, 0, 0});
    if (S > 1) {
        size_t blocks = ((size_t)M * N + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)ws, S, M, N, bias,
                           residual, Y, ldy, act);
    }
    return 0;
}

}  // namespace

// internal launcher shared with lstm.hip: Y[r, 0:N] (row stride ldy) = X[gather_idx[r], :] W^T + bias, r < min(M, *m_dev)
int mg_launch_linear(const float* X, int M, int K, const float* W, const float* bias, int N, float* Y, int ldy,
                     const int32_t* gather_idx, const int32_t* m_dev, hipStream_t stream) {
    return launch_gemm<false>(X, M, K, W, N, bias, nullptr, Y, ldy, MGNNS_ACT_NONE, gather_idx, m_dev, nullptr, 0, stream);
}

// internal strided-batch launcher (sq_mha_folded.hip): for z < nbatch
//   Y_z[m, 0:N] (row stride ldy) = X_z[m, 0:K] (row stride ldx) . W_z (+ bias_z),  P_z = P + z * stride_P
// W_z is [N,K] (w_is_kn == 0, the nn.Linear layout) or [K,N] (w_is_kn == 1).
int mg_launch_gemm_batched(const float* X, int ldx, long sx, int M, int K, const float* W, long sw, int w_is_kn,
                           const float* bias, long sb, int N, float* Y, int ldy, long sy, int nbatch,
                           hipStream_t stream) {
    const int vecX = (K % 4 == 0) && (ldx % 4 == 0) && (sx % 4 == 0) && mg_aligned16(X);
    const int vecW = (sw % 4 == 0) && mg_aligned16(W) && (w_is_kn ? (N % 4 == 0) : (K % 4 == 0));
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, nbatch);
    const GemmBatch bt{nbatch, sx, sw, sy, sb};
    if (use_small(M, N, K, ldx, X, W, w_is_kn != 0, sx, sw)) {
        dim3 gs((N + 63) / 64, (M + 15) / 16, nbatch);
        if (w_is_kn)
            hipLaunchKernelGGL(gemm_small_kernel<true>, gs, dim3(256), 0, stream, X, M, K, W, N, bias, nullptr, Y, ldy,
                               MGNNS_ACT_NONE, ldx, bt);
        else
            hipLaunchKernelGGL(gemm_small_kernel<false>, gs, dim3(256), 0, stream, X, M, K, W, N, bias, nullptr, Y, ldy,
                               MGNNS_ACT_NONE, ldx, bt);
        return 0;
    }
    if (w_is_kn)
        hipLaunchKernelGGL(gemm_f32_kernel<true>, grid, dim3(256), 0, stream, X, M, K, W, N, bias, nullptr, Y, ldy,
                           MGNNS_ACT_NONE, vecX, vecW, nullptr, nullptr, K, nullptr, ldx, bt);
    else
        hipLaunchKernelGGL(gemm_f32_kernel<false>, grid, dim3(256), 0, stream, X, M, K, W, N, bias, nullptr, Y, ldy,
                           MGNNS_ACT_NONE, vecX, vecW, nullptr, nullptr, K, nullptr, ldx, bt);
    return 0;
}

extern "C" size_t mgnns_gemm_workspace_bytes(void) { return (size_t)16 << 20; }   // 16 MiB of K-split partials

extern "C" int mgnns_linear_fwd(const float* X, int M, int K, const float* W, const float* bias, int N,
                                const float* residual, float* Y, int act, void* workspace, size_t workspace_bytes,
                                mgnns_stream_t stream) {
    MG_REQUIRE(X && W && Y, "mgnns_linear_fwd: null pointer");
    MG_REQUIRE(M >= 0 && K > 0 && N > 0, "mgnns_linear_fwd: bad dims M=%d K=%d N=%d", M, K, N);
    MG_REQUIRE(act >= 0 && act <= 2, "mgnns_linear_fwd: unknown activation %d", act);
    if (M == 0) return 0;
    launch_gemm<false>(X, M, K, W, N, bias, residual, Y, N, act, nullptr, nullptr, reinterpret_cast<float*>(workspace),
                       workspace ? workspace_bytes / sizeof(float) : 0, (hipStream_t)stream);
    MG_CHECK_LAUNCH("mgnns_linear_fwd");
    return 0;
}

extern "C" int mgnns_matmul_fwd(const float* X, int M, int K, const float* W, int N, float* Y, int act, void* workspace,
                                size_t workspace_bytes, mgnns_stream_t stream) {
    MG_REQUIRE(X && W && Y, "mgnns_matmul_fwd: null pointer");
    MG_REQUIRE(M >= 0 && K > 0 && N > 0, "mgnns_matmul_fwd: bad dims M=%d K=%d N=%d", M, K, N);
    MG_REQUIRE(act >= 0 && act <= 2, "mgnns_matmul_fwd: unknown activation %d", act);
    if (M == 0) return 0;
    launch_gemm<true>(X, M, K, W, N, nullptr, nullptr, Y, N, act, nullptr, nullptr, reinterpret_cast<float*>(workspace),
                      workspace ? workspace_bytes / sizeof(float) : 0, (hipStream_t)stream);
    MG_CHECK_LAUNCH("mgnns_matmul_fwd");
    return 0;
}
