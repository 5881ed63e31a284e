// Image memory bank + global max-pool in ONE pass over the [B,2048,196] feature map:
//   bank[b,p,:] = W * feat[b,:,p] + bias      (get_img_*_memory_bank, MODEL:400-428)
//   pooled[b,k] = max_p feat[b,k,p]           (MaxPool2d(14,14), MODEL:454-455)
// The reference permutes/copies the map to [B*196,2048] for the Linear and re-reads it for the
// pool; here the map is consumed in its native k-major layout (the A operand of the MFMA is read
// "transposed" straight out of LDS) and every element crosses HBM once.
//
// One 1024-thread workgroup per sample: C[208 x 304] = A[208 x K] * Wt[K x 304] on the exact-f32 MFMA
// (v_mfma_f32_16x16x4_f32), 13 x 19 tiles split over 16 waves as 4 (M) x 4 (N), BK = 16, double-
// buffered LDS with register prefetch of the next K-slice.
#include "common.hpp"

namespace {

constexpr int PT = 13, NT = 19;          // 16-wide tiles: P <= 208, N <= 304
constexpr int LDA = PT * 16;             // 208: LDS row stride of the A slice ([k][p]); 208 % 32 == 16
constexpr int LDB = NT * 16;             // 304: LDS row stride of the B slice ([k][n]); 304 % 32 == 16
constexpr int BK = 16;
constexpr int MT_W = 4, NT_W = 5;        // per-wave tile budget (4 x 4 wave grid)
constexpr int NTHR = 1024;
constexpr int NA = (BK * (LDA / 4) + NTHR - 1) / NTHR;   // float4 per thread per A slice (1)
constexpr int NB = (BK * (LDB / 4) + NTHR - 1) / NTHR;   // float4 per thread per B slice (2)

__global__ __launch_bounds__(NTHR) void imgbank_pool_kernel(const float* __restrict__ feat, int K, int P,
                                                           const float* __restrict__ Wt, int ldw,
                                                           const float* __restrict__ bias, int N,
                                                           float* __restrict__ bank, float* __restrict__ pooled) {
    __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int wm = wave >> 2, wn = wave & 3;             // 4 x 4 wave grid
    const int mt0 = wm == 0 ? 0 : 1 + 3 * wm, mtn = wm == 0 ? 4 : 3;    // 4 | 3 | 3 | 3 row tiles
    const int nt0 = wn * NT_W, ntn = wn == 3 ? NT - 3 * NT_W : NT_W;    // 5 5 5 | 4 col tiles
    const float* fb = feat + (size_t)b * K * P;
    const int P4 = P >> 2;
    const int nA4 = BK * P4;                             // float4 per A slice
    const int N4 = LDB / 4;                              // 76 float4 per B row
    const int nB4 = BK * N4;

    f32x4 acc[MT_W][NT_W];
#pragma unroll
    for (int i = 0; i < MT_W; ++i)
#pragma unroll
        for (int j = 0; j < NT_W; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 ra[NA], rb[NB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int q = tid + u * NTHR;
            ra[u] = q < nA4 ? reinterpret_cast<const f32x4*>(fb + (size_t)k0 * P)[q] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int q = tid + u * NTHR;
            if (q < nB4) {
                const int k = q / N4, c = q - k * N4;
                rb[u] = *reinterpret_cast<const f32x4*>(Wt + (size_t)(k0 + k) * ldw + 4 * c);
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int q = tid + u * NTHR;
            if (q < nA4) {
                const int k = q / P4, c = q - k * P4;
                *reinterpret_cast<f32x4*>(&As[buf][k * LDA + 4 * c]) = ra[u];
            }
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int q = tid + u * NTHR;
            if (q < nB4) *reinterpret_cast<f32x4*>(&Bs[buf][4 * q]) = rb[u];
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();
    const int nk = K / BK;
    for (int c = 0; c < nk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nk) gload((c + 1) * BK);
        // max-pool of this slice's 16 feature rows: wave w owns row w
        if (pooled) {
            const float* row = &As[buf][wave * LDA];
            float m = -INFINITY;
            for (int p = lane; p < P; p += 64) m = fmaxf(m, row[p]);
            m = wave_max(m);
            if (lane == 0) pooled[(size_t)b * K + c * BK + wave] = m;
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            const float* ap = &As[buf][(kk + (lane >> 4)) * LDA + mt0 * 16 + (lane & 15)];
            const float* bp = &Bs[buf][(kk + (lane >> 4)) * LDB + nt0 * 16 + (lane & 15)];
            float a[MT_W], bfr[NT_W];
#pragma unroll
            for (int i = 0; i < MT_W; ++i) a[i] = (i < mtn) ? ap[i * 16] : 0.f;
#pragma unroll
            for (int j = 0; j < NT_W; ++j) bfr[j] = (j < ntn) ? bp[j * 16] : 0.f;
#pragma unroll
            for (int i = 0; i < MT_W; ++i) {
                if (i < mtn) {
#pragma unroll
                    for (int j = 0; j < NT_W; ++j)
                        if (j < ntn) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bfr[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        if (c + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue: + bias, store bank[b, p, n]; C layout col = lane&15, row = (lane>>4)*4 + r
    float* ob = bank + (size_t)b * P * N;
#pragma unroll
    for (int j = 0; j < NT_W; ++j) {
        if (j >= ntn) continue;
        const int n = (nt0 + j) * 16 + (lane & 15);
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MT_W; ++i) {
            if (i >= mtn) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = (mt0 + i) * 16 + (lane >> 4) * 4 + r;
                if (p < P) ob[(size_t)p * N + n] = acc[i][j][r] + bv;
            }
        }
    }
}

}  // namespace

extern "C" int mgnns_imgbank_pool_fwd(const float* feat, int B, int K, int P, const float* Wt, int ldw,
                                      const float* bias, int N, float* bank, float* pooled,
                                      mgnns_stream_t stream) {
    MG_REQUIRE(feat && Wt && bank, "mgnns_imgbank_pool_fwd: null pointer");
    MG_REQUIRE(B >= 0 && K > 0 && K % BK == 0, "mgnns_imgbank_pool_fwd: K=%d must be a positive multiple of %d", K, BK);
    MG_REQUIRE(P > 0 && P <= LDA && P % 4 == 0, "mgnns_imgbank_pool_fwd: P=%d unsupported (multiple of 4, <= %d)", P, LDA);
    MG_REQUIRE(N > 0 && N <= LDB, "mgnns_imgbank_pool_fwd: N=%d unsupported (<= %d)", N, LDB);
    MG_REQUIRE(ldw >= LDB && ldw % 4 == 0, "mgnns_imgbank_pool_fwd: ldw=%d must be >= %d and a multiple of 4", ldw, LDB);
    MG_REQUIRE(mg_aligned16(feat) && mg_aligned16(Wt), "mgnns_imgbank_pool_fwd: feat/Wt must be 16-byte aligned");
    if (B == 0) return 0;
    hipLaunchKernelGGL(imgbank_pool_kernel, dim3(B), dim3(NTHR), 0, (hipStream_t)stream, feat, K, P, Wt, ldw, bias, N,
                       bank, pooled);
    MG_CHECK_LAUNCH("mgnns_imgbank_pool_fwd");
    return 0;
}
