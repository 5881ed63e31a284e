// bf16-operand image memory bank + global max-pool, one pass over the fp32 [B,2048,196] feature map
// (get_img_*_memory_bank, MODEL:400-428, fused with MaxPool2d(14,14), MODEL:454-455):
//   bank[b,p,:] = bf16( W * feat[b,:,p] + bias )   -> [B, P, 320] bf16 (model dim zero padded: the layout the
//                                                     bf16 fusion-attention kernel stages straight into LDS)
//   pooled[b,k] = max_p feat[b,k,p]                 (exact fp32)
// The kernel is HBM bound on the fp32 map (1.6 MB per sample-channel, read exactly once).
//
// Two 512-thread workgroups per sample, one per half of the 196 regions (7 | 6 row tiles of 16), each streams
// its half of every feature row (448-B segments) in BK = 128 slices: global -> registers (fp32; the max-pool is
// taken here) -> bf16 -> LDS transposed to [p][k] (XOR-swizzled 16-B chunks: conflict-free for both the
// ds_write_b128 of the transpose and the ds_read_b128 MFMA A fragments).  8 waves split the 19 column tiles
// (300 outputs) 3,3,3,2,2,2,2,2; the B operand (W) comes straight from L2 in a pre-packed fragment-major layout.
// v_mfma_f32_16x16x32_bf16, fp32 accumulation; epilogue transposes through LDS so bank rows leave as 16-B lanes.
#include "common.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifdef MG_IMG_TRACE
// profiling aid (off by default): per-phase cycle sums of wave 0 / wave 7 of two workgroups
__device__ unsigned long long g_img_trace[4][8];
#define IMG_T() __builtin_amdgcn_s_memtime()
#endif

namespace {

constexpr int BK = 128;                 // k-slice (4 MFMA k-steps): every lane of a wave streams 8 feature rows
constexpr int MTH = 7;                  // row tiles per half (112 rows)
constexpr int ROWS = MTH * 16;
constexpr int FSTR = 18;                // LDS row stride of the staged slice in 16-B chunks (16 data + 2 pad)
constexpr int NT = 19;                  // 304 / 16
constexpr int OUT_LD = 320;             // bank row length (bf16)
constexpr int OCH = OUT_LD / 8;         // 40 chunks per output row
constexpr int OSTR = OUT_LD * 2 + 16;   // epilogue LDS row stride in bytes (656: rows land on distinct banks)
constexpr int NTHR = 512;
constexpr int P_SPLIT = 104;            // half 0: regions [0,104) (tiles 0..6), half 1: [104,196) (tiles 0..5)

// two fp32 -> packed bf16x2 (round to nearest even) in ONE instruction; there is no builtin for it on gfx950
__device__ __forceinline__ unsigned int pack2(float a, float b) {
    unsigned int r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Wp[nt][ks][lane][8] = W[nt*16 + (lane&15)][ks*32 + (lane>>4)*8 + j]   (0 for rows >= N)
__global__ __launch_bounds__(256) void pack_w_kernel(const float* __restrict__ W, int N, int K, unsigned short* __restrict__ Wp) {
    const int KS = K / 32;
    const size_t total = (size_t)NT * KS * 64;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        size_t r = i >> 6;
        const int ks = (int)(r % KS);
        const int nt = (int)(r / KS);
        const int row = nt * 16 + (lane & 15);
        const int k0 = ks * 32 + (lane >> 4) * 8;
        uint4 o = make_uint4(0u, 0u, 0u, 0u);
        if (row < N) {
            const float* w = W + (size_t)row * K + k0;
            o.x = pack2(w[0], w[1]);
            o.y = pack2(w[2], w[3]);
            o.z = pack2(w[4], w[5]);
            o.w = pack2(w[6], w[7]);
        }
        reinterpret_cast<uint4*>(Wp)[i] = o;
    }
}

// NTN = column tiles of this wave (3 for waves 0-2, 2 for waves 3-7): compile-time so the MFMA stream is branch-free.
// Every wave runs the same number of barriers; the two instantiations only differ in the tile count.
template <int NTN>
__device__ __forceinline__ void imgbank_body(unsigned char* smem, const float* __restrict__ feat, int Bn, int K, int P,
                                             const unsigned short* __restrict__ Wp, const float* __restrict__ bias, int N,
                                             unsigned short* __restrict__ bank, float* __restrict__ pooled_part) {
    uint4* Fs = reinterpret_cast<uint4*>(smem);                            // [2][ROWS][FSTR] chunks
    uint4* Os = Fs + 2 * ROWS * FSTR;                                      // [ROWS][OCH] chunks (epilogue)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the two halves of a sample share the 128-B lines at their seam and the same W stream: keep the pair on ONE
    // XCD (workgroup id % 8 is the XCD in practice) so those lines are served by one L2; any mapping is correct
    const int xcd = blockIdx.x & 7, jq = blockIdx.x >> 3;
    const int b_raw = (jq >> 1) * 8 + xcd, mh = jq & 1;
    if (b_raw >= Bn) return;                   // tail when B % 8 != 0 (whole workgroup exits together)
    const int b = b_raw;
    const int p0 = mh ? P_SPLIT : 0;
    const int p_store_end = mh ? P : P_SPLIT;                              // rows [p0, p_store_end) are ours
    const int nt0 = wave < 3 ? 3 * wave : 9 + 2 * (wave - 3);
    const int KS = K / 32;

    // staging role: half-wave = 8-row group kc (0..15) of the 128-row slice, lane&31 = region quad pq
    const int pq = lane & 31, kc = 2 * wave + (lane >> 5);
    const int npq = mh ? (P - P_SPLIT + 3) / 4 : ROWS / 4;                 // 23 | 28 quads carry data
    const bool st_on = pq < ROWS / 4;                                      // lanes that own LDS rows (28 per half-wave)
    const bool ld_on = st_on && pq < npq && (p0 + 4 * pq + 3 < P);
    const float* fsrc = feat + ((size_t)b * K + 8 * kc) * P + p0 + 4 * pq;

    f32x4 acc[MTH][NTN];
#pragma unroll
    for (int i = 0; i < MTH; ++i)
#pragma unroll
        for (int j = 0; j < NTN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 st[8];
    auto gload = [&](int c) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            st[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            // streamed once: non-temporal, so the map does not evict the W fragments every workgroup re-reads from L2
            if (ld_on) st[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(fsrc + ((size_t)c * BK + i) * P));
        }
    };
    // A slice goes global -> st (fp32) -> pk (max-pool taken, packed bf16, half the registers) -> LDS.  Converting at
    // the TOP of an iteration frees st for the next slice's loads before the MFMAs start, so those loads have the
    // whole iteration (MFMAs + LDS write + barrier) to land and the memory pipe idles only during the conversion.
    uint4 pk[4];
    auto convert = [&](int c) {
        // max-pool of the 8 feature rows this wave just loaded (exact fp32): each half-wave = two DPP rows holds one
        // 8-row group: lanes 0/16 and 32/48 carry the row maxima; lane i keeps the result of feature row i of the
        // wave's 16, so they leave as ONE 64-byte store per wave
        float mine = -INFINITY;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float m = ld_on ? fmaxf(fmaxf(st[i][0], st[i][1]), fmaxf(st[i][2], st[i][3])) : -INFINITY;
            m = row16_max(m);
            const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 0));
            const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 16));
            const float b0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 32));
            const float b1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 48));
            if (lane == i) mine = fmaxf(a0, a1);
            if (lane == 8 + i) mine = fmaxf(b0, b1);
        }
        if (lane < 16) pooled_part[((size_t)b * 2 + mh) * K + c * BK + 16 * wave + lane] = mine;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pk[j].x = pack2(st[0][j], st[1][j]);
            pk[j].y = pack2(st[2][j], st[3][j]);
            pk[j].z = pack2(st[4][j], st[5][j]);
            pk[j].w = pack2(st[6][j], st[7][j]);
        }
    };
    auto lstore = [&](int buf) {          // transpose-write of the packed slice
        if (st_on) {
#pragma unroll
            for (int j = 0; j < 4; ++j) Fs[(buf * ROWS + 4 * pq + j) * FSTR + (kc ^ (pq & 7))] = pk[j];
        }
    };

    const uint4* wb = reinterpret_cast<const uint4*>(Wp) + (size_t)nt0 * KS * 64 + lane;
    const int nchunk = K / BK;
    // B fragments (L2) of ALL four k-steps of a slice are requested at the top of its iteration, BEFORE the next map
    // slice is requested: vector-memory results return in order, so a W fragment requested after the map loads
    // would not be usable until those (HBM latency) have landed -- that stall, once per slice, used to be ~60 % of
    // the MFMA phase
    uint4 bq[BK / 32][NTN];
    constexpr int HA = 4, HB = MTH - HA;
    uint4 ga[HA], gb[HB];
    auto afrag_a = [&](const uint4* fb, int kk) {
#pragma unroll
        for (int i = 0; i < HA; ++i) {
            const int row = i * 16 + (lane & 15);
            ga[i] = fb[row * FSTR + ((4 * kk + (lane >> 4)) ^ ((row >> 2) & 7))];
        }
    };
    auto afrag_b = [&](const uint4* fb, int kk) {
#pragma unroll
        for (int i = 0; i < HB; ++i) {
            const int row = (HA + i) * 16 + (lane & 15);
            gb[i] = fb[row * FSTR + ((4 * kk + (lane >> 4)) ^ ((row >> 2) & 7))];
        }
    };
    gload(0);
    convert(0);
    lstore(0);
    if (nchunk > 1) gload(1);
    __syncthreads();
#ifdef MG_IMG_TRACE
    unsigned long long tc = 0, tm = 0, tb = 0, t_start = IMG_T();
#endif
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
#ifdef MG_IMG_TRACE
        const unsigned long long t0 = IMG_T();
#endif
        const uint4* fb = Fs + (size_t)buf * ROWS * FSTR;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk)
#pragma unroll
            for (int j = 0; j < NTN; ++j) bq[kk][j] = wb[((size_t)j * KS + c * (BK / 32) + kk) * 64];
        afrag_a(fb, 0);                           // first k-step's fragments fly under the conversion below
        afrag_b(fb, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < nchunk) {
            convert(c + 1);                       // waits for slice c+1 (requested one iteration ago)
            if (c + 2 < nchunk) gload(c + 2);     // ... and its registers take slice c+2 at once
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef MG_IMG_TRACE
        const unsigned long long t1 = IMG_T();
#endif
        // MFMAs of the slice, software pipelined in two half-groups of row tiles: while one half's MFMAs run, the
        // other half's A fragments (ds_read_b128) for the same / next k-step are in flight, so an LDS round trip is
        // exposed once per slice (and that one overlaps the conversion above) instead of once per k-step
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < HA; ++i) {
                const bf16x8 av = __builtin_bit_cast(bf16x8, ga[i]);
#pragma unroll
                for (int j = 0; j < NTN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bq[kk][j]), av,
                                                                       acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 1 < BK / 32) afrag_a(fb, kk + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < HB; ++i) {
                const bf16x8 av = __builtin_bit_cast(bf16x8, gb[i]);
#pragma unroll
                for (int j = 0; j < NTN; ++j)
                    acc[HA + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bq[kk][j]), av,
                                                                            acc[HA + i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 1 < BK / 32) afrag_b(fb, kk + 1);
        }
#ifdef MG_IMG_TRACE
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t2 = IMG_T();
#endif
        if (c + 1 < nchunk) lstore(buf ^ 1);
        mg_lds_barrier();        // NOT __syncthreads(): its vmcnt(0) would wait for the map slice requested above
#ifdef MG_IMG_TRACE
        const unsigned long long t3 = IMG_T();
        tc += t1 - t0; tm += t2 - t1; tb += t3 - t2;
#endif
    }
#ifdef MG_IMG_TRACE
    if ((lane == 0) && (wave == 0 || wave == 7) && (blockIdx.x == 0 || blockIdx.x == 301)) {
        unsigned long long* g = g_img_trace[(blockIdx.x ? 2 : 0) + (wave ? 1 : 0)];
        g[0] = tc; g[1] = tm; g[2] = tb; g[3] = IMG_T() - t_start;
    }
#endif

    // ---- epilogue: + bias, bf16, through LDS, 16-B row stores; columns N..319 are zero -----------------------
    // The tiles were computed TRANSPOSED (A operand = W fragment, B operand = map fragment): this lane's accumulator
    // element [i][j][r] is output column (nt0+j)*16 + 4*(lane>>4) + r of region row 16i + (lane&15), i.e. four
    // CONSECUTIVE bank columns per tile -> one 8-byte LDS write (the untransposed layout needs four 2-byte writes).
    unsigned char* osb = reinterpret_cast<unsigned char*>(Os);
#pragma unroll
    for (int j = 0; j < NTN; ++j) {
        const int n = (nt0 + j) * 16 + (lane >> 4) * 4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = (n + r < N) ? bias[n + r] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < MTH; ++i) {
            const int row = i * 16 + (lane & 15);
            uint2 o;
            o.x = pack2(n + 0 < N ? acc[i][j][0] + bv[0] : 0.f, n + 1 < N ? acc[i][j][1] + bv[1] : 0.f);
            o.y = pack2(n + 2 < N ? acc[i][j][2] + bv[2] : 0.f, n + 3 < N ? acc[i][j][3] + bv[3] : 0.f);
            *reinterpret_cast<uint2*>(osb + (size_t)row * OSTR + n * 2) = o;
        }
    }
    // zero the pad columns 304..319 (two chunks per row)
    for (int q = tid; q < ROWS * 2; q += NTHR)
        *reinterpret_cast<uint4*>(osb + (size_t)(q >> 1) * OSTR + (NT * 2 + (q & 1)) * 16) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    uint4* ob = reinterpret_cast<uint4*>(bank) + (size_t)b * P * OCH;
    const int nrows = p_store_end - p0;
    for (int q = tid; q < nrows * OCH; q += NTHR) {
        const int row = q / OCH, ch = q - row * OCH;
        ob[(size_t)(p0 + row) * OCH + ch] = *reinterpret_cast<const uint4*>(osb + (size_t)row * OSTR + ch * 16);
    }
}

__global__ __launch_bounds__(NTHR) void imgbank_pool_bf16_kernel(const float* __restrict__ feat, int Bn, int K, int P,
                                                                 const unsigned short* __restrict__ Wp,
                                                                 const float* __restrict__ bias, int N,
                                                                 unsigned short* __restrict__ bank,
                                                                 float* __restrict__ pooled_part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) < 3 * 64)
        imgbank_body<3>(smem, feat, Bn, K, P, Wp, bias, N, bank, pooled_part);
    else
        imgbank_body<2>(smem, feat, Bn, K, P, Wp, bias, N, bank, pooled_part);
}

// pooled[b,k] = max(part[b,0,k], part[b,1,k])
__global__ __launch_bounds__(256) void pool_combine_kernel(const float* __restrict__ part, int B, int K, float* __restrict__ pooled) {
    const size_t total = (size_t)B * K;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t b = i / K, k = i - b * K;
        pooled[i] = fmaxf(part[(b * 2) * K + k], part[(b * 2 + 1) * K + k]);
    }
}

constexpr size_t SMEM_BYTES = (size_t)(2 * ROWS * FSTR) * 16 + (size_t)ROWS * OSTR;

}  // namespace

#ifdef MG_IMG_TRACE
extern "C" int mgnns_debug_img_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_img_trace), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : 1;
}
#endif

extern "C" size_t mgnns_imgbank_packed_weight_bytes(int K) { return (size_t)NT * (K / 32) * 64 * 16; }

extern "C" int mgnns_imgbank_pack_weights_bf16(const float* W, int N, int K, void* Wp, mgnns_stream_t stream) {
    MG_REQUIRE(W && Wp, "mgnns_imgbank_pack_weights_bf16: null pointer");
    MG_REQUIRE(N > 0 && N <= NT * 16 && K > 0 && K % BK == 0, "mgnns_imgbank_pack_weights_bf16: unsupported N=%d K=%d", N, K);
    const size_t total = (size_t)NT * (K / 32) * 64;
    hipLaunchKernelGGL(pack_w_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, N, K,
                       reinterpret_cast<unsigned short*>(Wp));
    MG_CHECK_LAUNCH("mgnns_imgbank_pack_weights_bf16");
    return 0;
}

extern "C" int mgnns_imgbank_pool_bf16_fwd(const float* feat, int B, int K, int P, const void* Wp, const float* bias, int N,
                                           void* bank_bf16, int ld, float* pooled, float* pooled_work,
                                           mgnns_stream_t stream) {
    MG_REQUIRE(feat && Wp && bank_bf16 && pooled_work, "mgnns_imgbank_pool_bf16_fwd: null pointer");
    MG_REQUIRE(B >= 0 && K > 0 && K % BK == 0, "mgnns_imgbank_pool_bf16_fwd: K=%d must be a positive multiple of %d", K, BK);
    MG_REQUIRE(P % 4 == 0 && P > P_SPLIT && P <= P_SPLIT + (MTH - 1) * 16,
               "mgnns_imgbank_pool_bf16_fwd: P=%d unsupported (multiple of 4 in (%d, %d])", P, P_SPLIT, P_SPLIT + (MTH - 1) * 16);
    MG_REQUIRE(N > 0 && N <= NT * 16, "mgnns_imgbank_pool_bf16_fwd: N=%d unsupported (<= %d)", N, NT * 16);
    MG_REQUIRE(ld == OUT_LD, "mgnns_imgbank_pool_bf16_fwd: bank row length must be %d", OUT_LD);
    MG_REQUIRE(mg_aligned16(feat) && mg_aligned16(Wp) && mg_aligned16(bank_bf16),
               "mgnns_imgbank_pool_bf16_fwd: feat/Wp/bank must be 16-byte aligned");
    if (B == 0) return 0;
    MG_DYN_LDS(imgbank_pool_bf16_kernel, SMEM_BYTES);
    hipStream_t s = (hipStream_t)stream;
    const int nblk = ((B + 7) / 8) * 16;      // pairs laid out XCD-major, padded to a multiple of 8 samples
    hipLaunchKernelGGL(imgbank_pool_bf16_kernel, dim3(nblk), dim3(NTHR), SMEM_BYTES, s, feat, B, K, P,
                       reinterpret_cast<const unsigned short*>(Wp), bias, N, reinterpret_cast<unsigned short*>(bank_bf16),
                       pooled_work);
    if (pooled) {                              // pooled == NULL: the caller consumes the two halves in pooled_work itself
        const size_t total = (size_t)B * K;
        size_t blocks = (total + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(pool_combine_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)pooled_work, B, K, pooled);
    }
    MG_CHECK_LAUNCH("mgnns_imgbank_pool_bf16_fwd");
    return 0;
}
