// The two-workgroups-per-sample form of the bf16 image bank + max-pool (rounds 1-2), kept for SMALL batches: a sample's map is
// split over two 512-thread workgroups (region halves), so B samples occupy 2 B compute units and a workgroup's chain is half
// as long -- below ~160 samples, where the one-workgroup-per-sample stream kernel (imgbank_bf16.hip) leaves most of the chip
// idle, this one is faster (B = 32: 0.43 vs 0.47 ms per forward).  At chip-filling batches it loses: 448-B half rows
// (1.19x sector over-fetch), map rows through registers (3.5 TB/s against 4.3).  Same operands, same results.
#include "common.hpp"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifdef MG_IMG_TRACE
// profiling aid (off by default): per-phase cycle sums of wave 0 / wave 7 of two workgroups
__device__ unsigned long long g_img_pairs_trace[4][8];
#define IMG_T() __builtin_amdgcn_s_memtime()
#endif

#ifndef MG_IMG_AUX
#define MG_IMG_AUX 2          // cache policy of the map loads (aux bits of buffer_load): 2 = non-temporal
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

// ---- roles ------------------------------------------------------------------------------------------------------------
// Waves 4-7 are PRODUCERS: they own the HBM stream.  Each keeps THREE k-slices of its share of the map in flight in
// registers (a quarter of the 128 feature rows x its 28 region quads: 16 x 16 B per lane and slice), takes the max-pool,
// converts to bf16 and transposes into the LDS double buffer.  With the loads issued by the same waves that also ran the
// MFMAs (round 1) only one slice could be in flight per CU and the memory pipe idled through every conversion:
// ~56 KB in flight per CU at ~4 us per slice = 3.5 TB/s.
// Waves 0-3 are CONSUMERS: MFMA only (7 row tiles x 5 | 5 | 5 | 4 column tiles), A fragments from LDS through a 3-deep
// register ring, B fragments (W, L2 resident) of the NEXT slice requested as soon as the current slice's MFMAs are issued,
// so they fly across the slice barrier.  One LDS-only barrier per slice orders the two groups.
struct ImgGeom {
    int b, mh, p0, p_store_end, K, P, KS, nchunk;
};

constexpr int PMS = 36;                 // row stride (floats) of a producer wave's partial-maxima tile [32 rows][28 -> 36]

__device__ __forceinline__ void imgbank_producer(uint4* __restrict__ Fs, float* __restrict__ s_pm_all, const float* __restrict__ feat,
                                                 const ImgGeom& gm, float* __restrict__ pooled_part, int pw, int lane) {
    float* s_pm = s_pm_all + pw * 32 * PMS;
    const int pq = lane & 31, hw = lane >> 5;
    const int kc0 = 4 * pw + 2 * hw;                                    // this half-wave's two 8-row groups: kc0, kc0 + 1
    const int npq = gm.mh ? (gm.P - P_SPLIT + 3) / 4 : ROWS / 4;        // 23 | 28 quads carry data
    const bool st_on = pq < ROWS / 4;                                   // lanes that own LDS rows (28 per half-wave)
    const bool ld_on = st_on && pq < npq && (gm.p0 + 4 * pq + 3 < gm.P);
    // The map is read through a buffer resource: ONE per-lane byte offset (a single VGPR for all 48 loads in flight) plus
    // a wave-uniform row offset in an SGPR.  With flat 64-bit per-load addresses the compiler recycled address registers
    // between the sets and put s_waitcnt vmcnt(31..39) in front of every refill, i.e. a refill waited for the NEXT set.
    // lanes without data (pq >= npq) load what lane npq - 1 loads and ignore it: the loads must be UNCONDITIONAL -- behind a
    // branch the compiler no longer knows how many are outstanding and every wait becomes vmcnt(0)
    const int pqc = pq < npq ? pq : npq - 1;
    const int loff = (int)((16 * hw * gm.P + gm.p0 + 4 * pqc) * sizeof(float));
    const __amdgpu_buffer_rsrc_t frsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(feat + (size_t)gm.b * gm.K * gm.P), 0, gm.K * gm.P * (int)sizeof(float), 0x00027000);
    const int urow0 = 32 * pw;                                           // first feature row of this wave inside a slice
    const int P = gm.P, nchunk = gm.nchunk;

    f32x4 st[3][16];
    auto gload = [&](f32x4 (&s)[16], int c) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            // streamed once: non-temporal (aux bit 1), so the map does not evict the W fragments every workgroup re-reads from L2
            s[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(frsrc, loff, (c * BK + urow0 + i) * P * (int)sizeof(float), MG_IMG_AUX));
        }
    };
    // slice (in registers) -> max-pool of its 16 feature rows + bf16 transpose-write into LDS buffer `buf`
    auto emit = [&](f32x4 (&s)[16], int c, int buf) {
#ifdef MG_IMG_NO_EMIT
#pragma unroll
        for (int gi = 0; gi < 16; ++gi) asm volatile("" ::"v"(s[gi]));
        return;
#endif
        // max-pool (exact fp32): in-lane over the lane's 4 regions, then across the half-wave's 28 lanes THROUGH LDS -- the
        // wave writes its 32 x 32 partial maxima, 32 of its lanes read a feature row each (7 x 16 B) and finish it.  (The
        // cross-lane form -- 4 DPP steps + 4 v_readlane per row -- was most of the ~5 k cycles per slice a producer spent here,
        // and a producer that computes is a producer that does not refill the memory pipe.)
#pragma unroll
        for (int gi = 0; gi < 16; ++gi)
            s_pm[(16 * hw + gi) * PMS + pq] = ld_on ? fmaxf(fmaxf(s[gi][0], s[gi][1]), fmaxf(s[gi][2], s[gi][3])) : -INFINITY;
        if (st_on) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // lanes past the half's last region quad hold a copy of the last quad: their LDS rows are never stored
                    uint4 pk;
                    pk.x = pack2(s[8 * g + 0][j], s[8 * g + 1][j]);
                    pk.y = pack2(s[8 * g + 2][j], s[8 * g + 3][j]);
                    pk.z = pack2(s[8 * g + 4][j], s[8 * g + 5][j]);
                    pk.w = pack2(s[8 * g + 6][j], s[8 * g + 7][j]);
                    Fs[(buf * ROWS + 4 * pq + j) * FSTR + ((kc0 + g) ^ (pq & 7))] = pk;
                }
        }
        if (lane < 32) {                                       // same wave wrote s_pm: LDS operations of a wave complete in order
            const f32x4* r4 = reinterpret_cast<const f32x4*>(s_pm + lane * PMS);
            f32x4 m = r4[0];
#pragma unroll
            for (int q = 1; q < 7; ++q) {
                const f32x4 v = r4[q];
                m = f32x4{fmaxf(m[0], v[0]), fmaxf(m[1], v[1]), fmaxf(m[2], v[2]), fmaxf(m[3], v[3])};
            }
            pooled_part[((size_t)gm.b * 2 + gm.mh) * gm.K + c * BK + 32 * pw + lane] = fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3]));
        }
    };
    gload(st[0], 0);
    if (nchunk > 1) gload(st[1], 1);
    if (nchunk > 2) gload(st[2], 2);
    emit(st[0], 0, 0);
    if (nchunk > 3) gload(st[0], 3);
    mg_lds_barrier();
    // iteration c: slice c + 1 leaves its registers (set (c+1) % 3), which take slice c + 4 at once.  The steady-state
    // loop is branch-free: with a conditional load in it the compiler's wait for "the loads of this set" has to assume
    // the fewest newer loads any path issued, i.e. it waits for the NEXT set too and the third slice in flight is lost.
#ifdef MG_IMG_TRACE
    unsigned long long t_emit = 0, t_issue = 0, t_bar = 0, t_first = 0;
#define MG_TT(v) const unsigned long long v = IMG_T(); __builtin_amdgcn_sched_barrier(0)
#else
#define MG_TT(v)
#endif
#define MG_PFULL(S, c)                                                      \
    do {                                                                    \
        MG_TT(ta_);                                                         \
        asm volatile("" ::"v"(st[S][0]));   /* wait for the set's FIRST row only */ \
        __builtin_amdgcn_sched_barrier(0);                                  \
        MG_TT(tb_);                                                         \
        emit(st[S], (c) + 1, ((c) + 1) & 1);                                \
        __builtin_amdgcn_sched_barrier(0);                                  \
        MG_TT(tc_);                                                         \
        gload(st[S], (c) + 4);                                              \
        __builtin_amdgcn_sched_barrier(0); /* nothing of the NEXT emit is hoisted between / above these loads: it would wait for the next set */ \
        MG_TT(td_);                                                         \
        mg_lds_barrier();                                                   \
        __builtin_amdgcn_sched_barrier(0);                                  \
        MG_TT(te_);                                                         \
        MG_TACC();                                                          \
    } while (0)
#define MG_PSTEP(S, c)                                                      \
    do {                                                                    \
        if ((c) < nchunk) {                                                 \
            if ((c) + 1 < nchunk) {                                         \
                emit(st[S], (c) + 1, ((c) + 1) & 1);                        \
                if ((c) + 4 < nchunk) gload(st[S], (c) + 4);                \
            }                                                               \
            mg_lds_barrier();                                               \
        }                                                                   \
    } while (0)
#ifdef MG_IMG_TRACE
#define MG_TACC() do { t_first += tb_ - ta_; t_emit += tc_ - tb_; t_issue += td_ - tc_; t_bar += te_ - td_; } while (0)
#else
#define MG_TACC() do { } while (0)
#endif
    int c = 0;
    for (; c + 6 < nchunk; c += 3) {
        MG_PFULL(1, c);
        MG_PFULL(2, c + 1);
        MG_PFULL(0, c + 2);
    }
    for (; c < nchunk; c += 3) {
        MG_PSTEP(1, c);
        MG_PSTEP(2, c + 1);
        MG_PSTEP(0, c + 2);
    }
#undef MG_PFULL
#ifdef MG_IMG_TRACE
    if (lane == 0 && pw == 0 && (blockIdx.x == 0 || blockIdx.x == 301)) {
        unsigned long long* g = g_img_pairs_trace[(blockIdx.x ? 2 : 0) + 1];
        g[0] = t_first; g[1] = t_emit; g[2] = t_issue; g[3] = t_bar;
    }
#endif
#undef MG_PSTEP
}

template <int NTN>
__device__ __forceinline__ void imgbank_consumer(unsigned char* smem, const ImgGeom& gm, const unsigned short* __restrict__ Wp,
                                                 const float* __restrict__ bias, int N, int wave, int lane) {
    const uint4* Fs = reinterpret_cast<const uint4*>(smem);
    uint4* Os = reinterpret_cast<uint4*>(smem) + 2 * ROWS * FSTR;
    const int nt0 = 5 * wave, KS = gm.KS, nchunk = gm.nchunk;
    f32x4 acc[MTH][NTN];
#pragma unroll
    for (int i = 0; i < MTH; ++i)
#pragma unroll
        for (int j = 0; j < NTN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint4* wu = reinterpret_cast<const uint4*>(Wp) + (size_t)nt0 * KS * 64;      // wave-uniform: SGPR base + lane*16
    // B fragments: a ring of THREE k-steps (the four of a slice at once do not fit beside 140 accumulators); k-step g of
    // the whole K (= 4 c + kk) lives in ring[g % 3] and is requested when k-step g - 3 has issued its MFMAs, i.e. two
    // k-steps (70 MFMAs, ~0.5 us) ahead, across slice barriers too.  A consumer wave is alone on its SIMD's MFMA pipe, so what
    // the ring does not cover of the L2 latency is exposed -- the consumers have slack for it: a slice is ~1 us of MFMAs
    // against >= 3 us of HBM time.  (With a 2-deep ring they did not: 5.3 us per slice, 171 us per launch.)
    constexpr int RD = 3;
    uint4 bq[RD][NTN];
    const int nks = nchunk * (BK / 32);
    // A fragment of (k-step kk, row tile i): row = 16 i + (lane & 15), chunk = (4 kk + (lane >> 4)) ^ ((row >> 2) & 7).
    // (row >> 2) & 7 = (lane & 15) >> 2 for even i, + 4 for odd i, so the address is ONE per-lane base plus a
    // compile-time offset per (kk, i): no per-fragment address registers.
    const int abase_l = (lane & 15) * FSTR + ((lane >> 4) ^ ((lane & 15) >> 2));
    auto afrag = [&](const uint4* fb, int kk, int i) { return fb[abase_l + i * 16 * FSTR + ((4 * kk) ^ (4 * (i & 1)))]; };
#pragma unroll
    for (int d = 0; d < RD; ++d)
#pragma unroll
        for (int j = 0; j < NTN; ++j) bq[d][j] = (wu + ((size_t)j * KS + min(d, nks - 1)) * 64)[lane];
    mg_lds_barrier();
    // 12 k-steps (three slices) per trip so that ring slots are compile-time: K / 32 is a multiple of 4, the trip handles
    // slices c, c+1, c+2 with slot = (4 (c % 3) + kk) % 3
#ifdef MG_IMG_TRACE
    unsigned long long t_work = 0, t_cbar = 0;
#endif
    auto slice = [&](int c, auto slot0) {
        constexpr int S0 = decltype(slot0)::value;
        MG_TT(ta_);
        const uint4* fb = Fs + (size_t)(c & 1) * ROWS * FSTR;
        constexpr int AD = 6;                                    // A ring depth: fragments AD - 1 (k-step, row tile) pairs ahead
        uint4 ar[AD];
#pragma unroll
        for (int d = 0; d < AD - 1; ++d) ar[d] = afrag(fb, d / MTH, d % MTH);
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            constexpr int dummy = 0;
            (void)dummy;
#pragma unroll
            for (int i = 0; i < MTH; ++i) {
                const int sidx = kk * MTH + i;                       // A ring over the (k-step, row tile) pairs: a consumer wave is
                // alone on its SIMD, nothing else hides an LDS round trip (~130 cycles idle, more behind the producers' writes)
                if (sidx + AD - 1 < (BK / 32) * MTH)
                    ar[(sidx + AD - 1) % AD] = afrag(fb, (sidx + AD - 1) / MTH, (sidx + AD - 1) % MTH);
                const bf16x8 av = __builtin_bit_cast(bf16x8, ar[sidx % AD]);
#ifndef MG_IMG_NO_MFMA
#pragma unroll
                for (int j = 0; j < NTN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bq[(S0 + kk) % RD][j]), av,
                                                                       acc[i][j], 0, 0, 0);
#else
                asm volatile("" ::"v"(av));
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
            // unconditional (the last RD requests re-read the last k-step and are never used): behind a branch the
            // compiler cannot count the outstanding loads and every wait for a ring slot becomes vmcnt(0)
            const int gk = min(c * (BK / 32) + kk + RD, nks - 1);
#ifndef MG_IMG_NO_W
#pragma unroll
            for (int j = 0; j < NTN; ++j) bq[(S0 + kk) % RD][j] = (wu + ((size_t)j * KS + gk) * 64)[lane];
#else
            (void)gk;
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        MG_TT(tb_);
        mg_lds_barrier();
        __builtin_amdgcn_sched_barrier(0);
        MG_TT(tc_);
#ifdef MG_IMG_TRACE
        t_work += tb_ - ta_; t_cbar += tc_ - tb_;
#endif
    };
    for (int c = 0; c < nchunk; c += 3) {
        slice(c, std::integral_constant<int, 0>{});
        if (c + 1 < nchunk) slice(c + 1, std::integral_constant<int, (BK / 32) % RD>{});
        if (c + 2 < nchunk) slice(c + 2, std::integral_constant<int, (2 * (BK / 32)) % RD>{});
    }
#ifdef MG_IMG_TRACE
    if (lane == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == 301)) {
        unsigned long long* g = g_img_pairs_trace[(blockIdx.x ? 2 : 0)];
        g[0] = t_work; g[1] = t_cbar;
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
}

__global__ __launch_bounds__(NTHR) void imgbank_pool_bf16_pairs_kernel(const float* __restrict__ feat, int Bn, int K, int P,
                                                                 const unsigned short* __restrict__ Wp,
                                                                 const float* __restrict__ bias, int N,
                                                                 unsigned short* __restrict__ bank,
                                                                 float* __restrict__ pooled_part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* Fs = reinterpret_cast<uint4*>(smem);                            // [2][ROWS][FSTR] chunks
    unsigned char* osb = smem + (size_t)2 * ROWS * FSTR * 16;              // [ROWS][OSTR] bytes (epilogue)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the two halves of a sample share the 128-B lines at their seam and the same W stream: keep the pair on ONE
    // XCD (workgroup id % 8 is the XCD in practice) so those lines are served by one L2; any mapping is correct
    const int xcd = blockIdx.x & 7, jq = blockIdx.x >> 3;
    ImgGeom gm;
    gm.b = (jq >> 1) * 8 + xcd;
    gm.mh = jq & 1;
    if (gm.b >= Bn) return;                    // tail when B % 8 != 0 (whole workgroup exits together)
    gm.p0 = gm.mh ? P_SPLIT : 0;
    gm.p_store_end = gm.mh ? P : P_SPLIT;      // rows [p0, p_store_end) are ours
    gm.K = K; gm.P = P; gm.KS = K / 32; gm.nchunk = K / BK;

    float* s_pm_all = reinterpret_cast<float*>(osb + (size_t)ROWS * OSTR);      // [4][32][PMS]
    if (wave >= 4) imgbank_producer(Fs, s_pm_all, feat, gm, pooled_part, wave - 4, lane);
    else if (wave == 3) imgbank_consumer<4>(smem, gm, Wp, bias, N, wave, lane);
    else imgbank_consumer<5>(smem, gm, Wp, bias, N, wave, lane);

    // zero the pad columns 304..319 (two chunks per row), then the rows leave as 16-B lanes (all eight waves)
    for (int q = tid; q < ROWS * 2; q += NTHR)
        *reinterpret_cast<uint4*>(osb + (size_t)(q >> 1) * OSTR + (NT * 2 + (q & 1)) * 16) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    uint4* ob = reinterpret_cast<uint4*>(bank) + (size_t)gm.b * P * OCH;
    const int nrows = gm.p_store_end - gm.p0;
    for (int q = tid; q < nrows * OCH; q += NTHR) {
        const int row = q / OCH, ch = q - row * OCH;
        ob[(size_t)(gm.p0 + row) * OCH + ch] = *reinterpret_cast<const uint4*>(osb + (size_t)row * OSTR + ch * 16);
    }
}

// pooled[b,k] = max(part[b,0,k], part[b,1,k])
__global__ __launch_bounds__(256) void pool_combine_kernel(const float* __restrict__ part, int B, int K, float* __restrict__ pooled) {
    const size_t total = (size_t)B * K;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t b = i / K, k = i - b * K;
        pooled[i] = fmaxf(part[(b * 2) * K + k], part[(b * 2 + 1) * K + k]);
    }
}

constexpr size_t SMEM_BYTES = (size_t)(2 * ROWS * FSTR) * 16 + (size_t)ROWS * OSTR + (size_t)4 * 32 * PMS * sizeof(float);
static_assert(SMEM_BYTES <= 160 * 1024, "LDS");

}  // namespace

// (internal) launcher of the two-workgroups-per-sample form; arguments checked by mgnns_imgbank_pool_bf16_fwd
int mg_imgbank_pool_bf16_pairs(const float* feat, int B, int K, int P, const void* Wp, const float* bias, int N,
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
    MG_DYN_LDS(imgbank_pool_bf16_pairs_kernel, SMEM_BYTES);
    hipStream_t s = (hipStream_t)stream;
    const int nblk = ((B + 7) / 8) * 16;      // pairs laid out XCD-major, padded to a multiple of 8 samples
    hipLaunchKernelGGL(imgbank_pool_bf16_pairs_kernel, dim3(nblk), dim3(NTHR), SMEM_BYTES, s, feat, B, K, P,
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
