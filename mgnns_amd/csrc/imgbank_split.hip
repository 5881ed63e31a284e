// Split-bf16 ("bf16x3") image memory bank + global max-pool: the fp32-class form of csrc/imgbank_bf16.hip.
//   bank[b,p,:] = W * feat[b,:,p] + bias      (get_img_*_memory_bank, MODEL:400-428)  -> [B, P, N] fp32
//   pooled[b,h,k] = max over the regions of half h of feat[b,k,p]   (MaxPool2d(14,14), MODEL:454-455; exact fp32, two partial
//                   maxima per feature row: the fused channel tail takes their max while staging, anyone else amax(dim=1))
// Every fp32 operand x is carried as hi = bf16(x), lo = bf16(x - hi) and a product is formed as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi
// on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: ~2^-16 relative per product instead of bf16's 2^-8, at three MFMAs of
// the bf16 rate (the exact-f32 MFMA runs at 1/16 of it): the parity-grade mode's bank in ~1/4 of the exact kernel's time.
//
// Two 512-thread workgroups per sample, one per half of the 196 regions (112 | 84 rows = 7 | 6 row tiles of 16).  A BK = 64
// slice of the half's feature rows is loaded k-major as it lies in memory (wave w streams feature rows 8w..8w+7 of the slice,
// lane = one region quad: 8 x 16 B in flight per lane, requested a slice ahead), split and transposed in registers, and
// written to LDS as 16-B chunks of 8 consecutive k per region row (chunk index XOR row tile: conflict-free for the transposing
// ds_write_b128 and for the MFMA A-fragment ds_read_b128); wave w owns column tiles w, w + 8, w + 16 (3,3,3,2,2,2,2,2 of the
// 19) for all row tiles; the W fragments (hi, lo; mgnns_pack_weight_bf16_split layout, L2 resident) of the next k-step are
// requested before the current k-step's MFMAs.
#include "common.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int IS_BK = 64;                // k per slice: 2 MFMA k-steps, 8 chunks of 8
constexpr int IS_MT = 7;                 // row tiles per half (<= 112 rows)
constexpr int IS_ROWS = IS_MT * 16;
constexpr int IS_STR = 9;                // 16-B chunks per LDS row (8 data + 1 pad: rows r and r + 1 start 4 banks apart mod 64 ... 36 dwords)
constexpr int IS_NT = 19;                // column tiles (N <= 304)
constexpr int IS_TPW = 3;                // column tiles per wave (w, w + 8, w + 16)
constexpr int IS_THR = 512;
constexpr int IS_PSPLIT = 112;           // half 0: regions [0, 112), half 1: [112, P)
constexpr int IS_PMS = 29;               // row stride of the partial-maxima tile [64][28 -> 29]

__device__ __forceinline__ unsigned is_pack2(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__global__ __launch_bounds__(IS_THR) void imgbank_split_kernel(const float* __restrict__ feat, int K, int P,
                                                              const uint4* __restrict__ Wh, const uint4* __restrict__ Wl,
                                                              const float* __restrict__ bias, int N,
                                                              float* __restrict__ bank, float* __restrict__ pooled) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* Ahi = reinterpret_cast<uint4*>(smem);                                   // [2][IS_ROWS][IS_STR]
    uint4* Alo = Ahi + 2 * IS_ROWS * IS_STR;
    float* pm = reinterpret_cast<float*>(Alo + 2 * IS_ROWS * IS_STR);              // [2][64][IS_PMS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x >> 1, mh = blockIdx.x & 1;
    const int p0 = mh ? IS_PSPLIT : 0;
    const int rows = mh ? P - IS_PSPLIT : (P < IS_PSPLIT ? P : IS_PSPLIT);        // valid region rows of this half
    if (rows <= 0) {                                              // P <= 112: the second half is empty
        if (pooled)
            for (int k = tid; k < K; k += IS_THR) pooled[((size_t)b * 2 + mh) * K + k] = -INFINITY;
        return;
    }
    const int mtn = (rows + 15) / 16;                                              // row tiles with data
    const int nq = rows / 4;                                                        // region quads (P % 4 == 0)
    const int KS = K / 32, nk = K / IS_BK;
    const float* fb = feat + (size_t)b * K * P + p0;

    f32x4 acc[IS_MT][IS_TPW];
#pragma unroll
    for (int i = 0; i < IS_MT; ++i)
#pragma unroll
        for (int t = 0; t < IS_TPW; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- producer side (all waves): wave = chunk (8 feature rows of the slice), lane = region quad ----
    const int pq = lane < nq ? lane : nq - 1;                   // idle lanes repeat the last quad (unconditional loads), never store
    const bool st_on = lane < nq;
    f32x4 sl[8];
    auto gload = [&](int c) {
#pragma unroll
        for (int i = 0; i < 8; ++i) sl[i] = *reinterpret_cast<const f32x4*>(fb + (size_t)(c * IS_BK + wave * 8 + i) * P + 4 * pq);
    };
    auto emit = [&](int buf) {
        float* pmb = pm + buf * 64 * IS_PMS;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (st_on) pmb[(wave * 8 + i) * IS_PMS + lane] = fmaxf(fmaxf(sl[i][0], sl[i][1]), fmaxf(sl[i][2], sl[i][3]));
        if (st_on) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float h[8], l[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float x = sl[i][j];
                    h[i] = __builtin_bit_cast(float, (is_pack2(x, 0.f) << 16));          // bf16(x) as fp32
                    l[i] = x - h[i];
                }
                uint4 ch, cl;
                ch.x = is_pack2(h[0], h[1]); ch.y = is_pack2(h[2], h[3]); ch.z = is_pack2(h[4], h[5]); ch.w = is_pack2(h[6], h[7]);
                cl.x = is_pack2(l[0], l[1]); cl.y = is_pack2(l[2], l[3]); cl.z = is_pack2(l[4], l[5]); cl.w = is_pack2(l[6], l[7]);
                const int row = 4 * lane + j;
                const int at = (buf * IS_ROWS + row) * IS_STR + (wave ^ ((row >> 4) & 7));
                Ahi[at] = ch;
                Alo[at] = cl;
            }
        }
    };
    // ---- W fragments of this wave's column tiles, one k-step ahead ----
    size_t woff[IS_TPW];
    bool ton[IS_TPW];
#pragma unroll
    for (int t = 0; t < IS_TPW; ++t) {
        const int nt = wave + 8 * t;
        ton[t] = nt < IS_NT && nt * 16 < N;
        woff[t] = (size_t)(ton[t] ? nt : 0) * KS * 64 + lane;
    }
    uint4 bh[2][IS_TPW], bl[2][IS_TPW];
    auto wload = [&](int ks, int slot) {
#pragma unroll
        for (int t = 0; t < IS_TPW; ++t) {
            bh[slot][t] = Wh[woff[t] + (size_t)ks * 64];
            bl[slot][t] = Wl[woff[t] + (size_t)ks * 64];
        }
    };
    const int fr = lane & 15, fg = lane >> 4;

    gload(0);
    wload(0, 0);
    emit(0);
    __syncthreads();
    for (int c = 0; c < nk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nk) gload(c + 1);
        // the slice's partial maxima (written before the barrier that opened this iteration) -> pooled, 64 threads
        if (pooled && tid < 64) {
            const float* r = pm + buf * 64 * IS_PMS + tid * IS_PMS;
            float m = -INFINITY;
            for (int q = 0; q < nq; ++q) m = fmaxf(m, r[q]);
            pooled[((size_t)b * 2 + mh) * K + c * IS_BK + tid] = m;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ks = c * 2 + s;
            if (ks + 1 < KS) wload(ks + 1, (s + 1) & 1);
#pragma unroll
            for (int i = 0; i < IS_MT; ++i) {
                if (i < mtn) {
                    const int at = (buf * IS_ROWS + i * 16 + fr) * IS_STR + ((4 * s + fg) ^ (i & 7));
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, Ahi[at]);
                    const bf16x8 al = __builtin_bit_cast(bf16x8, Alo[at]);
#pragma unroll
                    for (int t = 0; t < IS_TPW; ++t) {
                        if (ton[t]) {
                            const bf16x8 wh = __builtin_bit_cast(bf16x8, bh[s & 1][t]), wl = __builtin_bit_cast(bf16x8, bl[s & 1][t]);
                            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wh, acc[i][t], 0, 0, 0);
                            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wl, acc[i][t], 0, 0, 0);
                            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wh, acc[i][t], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (c + 1 < nk) emit(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: + bias; acc[i][t][r] = bank[p0 + 16 i + 4 (lane >> 4) + r][16 nt + (lane & 15)] ----
    float* ob = bank + ((size_t)b * P + p0) * N;
#pragma unroll
    for (int t = 0; t < IS_TPW; ++t) {
        const int n = (wave + 8 * t) * 16 + fr;
        if (!ton[t] || n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < IS_MT; ++i) {
            if (i >= mtn) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = i * 16 + fg * 4 + r;
                if (p < rows) ob[(size_t)p * N + n] = acc[i][t][r] + bv;
            }
        }
    }
}

}  // namespace

extern "C" int mgnns_imgbank_pool_split_fwd(const float* feat, int B, int K, int P, const void* Wp_hi, const void* Wp_lo,
                                            const float* bias, int N, float* bank, float* pooled_halves,
                                            mgnns_stream_t stream) {
    MG_REQUIRE(feat && Wp_hi && Wp_lo && bank, "mgnns_imgbank_pool_split_fwd: null pointer");
    MG_REQUIRE(B >= 0 && K > 0 && K % IS_BK == 0, "mgnns_imgbank_pool_split_fwd: K=%d must be a positive multiple of %d", K, IS_BK);
    MG_REQUIRE(P > 0 && P % 4 == 0 && P <= IS_PSPLIT + IS_ROWS, "mgnns_imgbank_pool_split_fwd: P=%d unsupported (multiple of 4, <= %d)", P,
               IS_PSPLIT + IS_ROWS);
    MG_REQUIRE(N > 0 && N <= IS_NT * 16, "mgnns_imgbank_pool_split_fwd: N=%d unsupported (<= %d)", N, IS_NT * 16);
    MG_REQUIRE(mg_aligned16(feat) && mg_aligned16(Wp_hi) && mg_aligned16(Wp_lo), "mgnns_imgbank_pool_split_fwd: feat / weights must be 16-byte aligned");
    if (B == 0) return 0;
    const size_t lds = (size_t)4 * IS_ROWS * IS_STR * 16 + (size_t)2 * 64 * IS_PMS * sizeof(float);
    MG_DYN_LDS(imgbank_split_kernel, lds);
    hipLaunchKernelGGL(imgbank_split_kernel, dim3(2 * B), dim3(IS_THR), lds, (hipStream_t)stream, feat, K, P,
                       reinterpret_cast<const uint4*>(Wp_hi), reinterpret_cast<const uint4*>(Wp_lo), bias, N, bank, pooled_halves);
    MG_CHECK_LAUNCH("mgnns_imgbank_pool_split_fwd");
    return 0;
}
