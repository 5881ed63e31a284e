// Generic fp32 GEMM on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32): the small dense layers of
// the path (nn.Linear / Conv1d(k=1) / GraphConvolution X*W / read-out).  64x64 block tile,
// 4 waves as 2x2, each wave 2x2 MFMA tiles of 16x16, BK = 16, LDS-staged operands.
//   NT: Y = act(X[M,K] * W[N,K]^T + bias) + residual      (nn.Linear weight layout)
//   NN: Y = act(X[M,K] * W[K,N])                          (GraphConvolution weight layout)
#include "common.hpp"

namespace {

constexpr int BM = 64, BN = 64, BK = 16;
constexpr int SA = BK + 2;    // [row][k] stride: 18 floats -> conflict-free ds_read_b32 fragments
constexpr int SBN = BN + 16;  // [k][n] stride for the NN form: 80 floats

// load 4 consecutive floats p[0..3] with element-wise bound `n_valid` (<=4), vector path if aligned
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

template <bool W_IS_KN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ X, int M, int K,
                                                       const float* __restrict__ W, int N,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ residual,
                                                       float* __restrict__ Y, int ldy, int act, int vecX, int vecW,
                                                       const int32_t* __restrict__ gather_idx,
                                                       const int32_t* __restrict__ m_dev) {
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

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging map: 64 rows x 4 float4 per BK-slice = 256 float4, one per thread
    const int s_row = tid >> 2, s_k4 = (tid & 3) * 4;
    // optional row gather on the A side: logical row r reads X[gather_idx[r], :]
    const size_t a_row = (m0 + s_row < M) ? (gather_idx ? (size_t)gather_idx[m0 + s_row] : (size_t)(m0 + s_row)) : 0;

    for (int k0 = 0; k0 < K; k0 += BK) {
        {   // A tile
            const int gm = m0 + s_row, gk = k0 + s_k4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gm < M) v = load4(X + a_row * K + gk, K - gk, vecX);
            float* d = &As[s_row * SA + s_k4];
            *reinterpret_cast<float2*>(d) = float2{v[0], v[1]};
            *reinterpret_cast<float2*>(d + 2) = float2{v[2], v[3]};
        }
        if (!W_IS_KN) {  // W [N,K]
            const int gn = n0 + s_row, gk = k0 + s_k4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gn < N) v = load4(W + (size_t)gn * K + gk, K - gk, vecW);
            float* d = &Bs[s_row * SA + s_k4];
            *reinterpret_cast<float2*>(d) = float2{v[0], v[1]};
            *reinterpret_cast<float2*>(d + 2) = float2{v[2], v[3]};
        } else {         // W [K,N]: 16 k-rows x 16 float4
            const int kr = tid >> 4, n4 = (tid & 15) * 4;
            const int gk = k0 + kr, gn = n0 + n4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gk < K) v = load4(W + (size_t)gk * N + gn, N - gn, vecW);
            *reinterpret_cast<f32x4*>(&Bs[kr * SBN + n4]) = v;
        }
        __syncthreads();
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
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gn = n0 + wc * 32 + j * 16 + (lane & 15);
            if (gn >= N) continue;
            const float bv = bias ? bias[gn] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wr * 32 + i * 16 + (lane >> 4) * 4 + r;
                if (gm >= M) continue;
                float v = mg_act(acc[i][j][r] + bv, act);
                if (residual) v += residual[(size_t)gm * N + gn];
                Y[(size_t)gm * ldy + gn] = v;
            }
        }
}

}  // namespace

// internal launcher shared with lstm.hip: Y[r, 0:N] (row stride ldy) = X[gather_idx[r], :] W^T + bias, r < min(M, *m_dev)
int mg_launch_linear(const float* X, int M, int K, const float* W, const float* bias, int N, float* Y, int ldy,
                     const int32_t* gather_idx, const int32_t* m_dev, hipStream_t stream) {
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
    const int vecX = (K % 4 == 0) && mg_aligned16(X);
    const int vecW = (K % 4 == 0) && mg_aligned16(W);
    hipLaunchKernelGGL(gemm_f32_kernel<false>, grid, dim3(256), 0, stream, X, M, K, W, N, bias,
                       (const float*)nullptr, Y, ldy, MGNNS_ACT_NONE, vecX, vecW, gather_idx, m_dev);
    return 0;
}

extern "C" int mgnns_linear_fwd(const float* X, int M, int K, const float* W, const float* bias, int N,
                                const float* residual, float* Y, int act, mgnns_stream_t stream) {
    MG_REQUIRE(X && W && Y, "mgnns_linear_fwd: null pointer");
    MG_REQUIRE(M >= 0 && K > 0 && N > 0, "mgnns_linear_fwd: bad dims M=%d K=%d N=%d", M, K, N);
    MG_REQUIRE(act >= 0 && act <= 2, "mgnns_linear_fwd: unknown activation %d", act);
    if (M == 0) return 0;
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
    const int vecX = (K % 4 == 0) && mg_aligned16(X);
    const int vecW = (K % 4 == 0) && mg_aligned16(W);
    hipLaunchKernelGGL(gemm_f32_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, X, M, K, W, N, bias,
                       residual, Y, N, act, vecX, vecW, (const int32_t*)nullptr, (const int32_t*)nullptr);
    MG_CHECK_LAUNCH("mgnns_linear_fwd");
    return 0;
}

extern "C" int mgnns_matmul_fwd(const float* X, int M, int K, const float* W, int N, float* Y, int act,
                                mgnns_stream_t stream) {
    MG_REQUIRE(X && W && Y, "mgnns_matmul_fwd: null pointer");
    MG_REQUIRE(M >= 0 && K > 0 && N > 0, "mgnns_matmul_fwd: bad dims M=%d K=%d N=%d", M, K, N);
    MG_REQUIRE(act >= 0 && act <= 2, "mgnns_matmul_fwd: unknown activation %d", act);
    if (M == 0) return 0;
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
    const int vecX = (K % 4 == 0) && mg_aligned16(X);
    const int vecW = (N % 4 == 0) && mg_aligned16(W);
    hipLaunchKernelGGL(gemm_f32_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, X, M, K, W, N,
                       (const float*)nullptr, (const float*)nullptr, Y, N, act, vecX, vecW, (const int32_t*)nullptr,
                       (const int32_t*)nullptr);
    MG_CHECK_LAUNCH("mgnns_matmul_fwd");
    return 0;
}
