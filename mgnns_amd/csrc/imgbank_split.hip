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
// a 32-lane half of the wave four of them, lane & 31 = one region quad: 4 x 16 B in flight per lane, requested a slice ahead), split and
// transposed in registers (the halves exchange their half chunks through v_permlane32_swap), and
// written to LDS as 16-B chunks of 8 consecutive k per region row (chunk index XOR row tile: conflict-free for the transposing
// ds_write_b128 and for the MFMA A-fragment ds_read_b128); wave w owns column tiles w, w + 8, w + 16 (3,3,3,2,2,2,2,2 of the
// 19) for all row tiles; the W fragments (hi, lo; mgnns_pack_weight_bf16_split layout, L2 resident) of the next k-step are
// ring of three k-steps: requested TWO k-steps (~2000 cycles of MFMAs) before their use -- round 4's one k-step ahead left every
// k-step waiting for an L2 round trip (312 us per launch, 24 % of the pipe).
// Outputs (round 5): the fp32 bank and / or its split-bf16 images hi = bf16(x), lo = bf16(x - hi) as [B, P, 320] bf16 each (zero
// padded) -- the operand of the split-bf16 attention core (sq_mha_split_bf16.hip), written here instead of by a conversion pass.
#include "common.hpp"
#include "sq_mha_util.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifdef MG_IS_TRACE
// profiling aid (off by default; tools/dev/is_trace.py): s_memtime sums of a slice's phases, waves 0 (converts first) and 4
// (multiplies first) of workgroup 0
__device__ unsigned long long g_is_trace[2][8];
#define IS_T(i) { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); tt[i] += t1_ - t0_; t0_ = t1_; }
#else
#define IS_T(i)
#endif

namespace {

constexpr int IS_BK = 64;                // k per slice: 2 MFMA k-steps, 8 chunks of 8
constexpr int IS_MT = 7;                 // row tiles per half (<= 112 rows)
constexpr int IS_ROWS = IS_MT * 16;
constexpr int IS_STR = 9;                // 16-B chunks per LDS row (8 data + 1 pad: rows r and r + 1 start 4 banks apart mod 64 ... 36 dwords)
constexpr int IS_NT = 19;                // column tiles (N <= 304)
constexpr int IS_TPW = 3;                // column tiles per wave (w, w + 8, w + 16)
constexpr int IS_THR = 512;
constexpr int IS_PSPLIT = 112;           // half 0: regions [0, 112), half 1: [112, P)

__device__ __forceinline__ unsigned is_pack2(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

constexpr int IS_WD = 3;                 // W-fragment ring: two k-steps in flight + the one being multiplied (four: 21 spilled registers)
constexpr int IS_LD = 320;               // row length of the split-bf16 bank images
#ifndef MG_IS_AD
#define MG_IS_AD 2
#endif
#ifndef MG_IS_PF
#define MG_IS_PF 1
#endif
constexpr int IS_PF = MG_IS_PF;          // slices of feature rows in flight ahead of their conversion (1 = rounds 3-5)
static_assert(IS_PF == 1 || IS_PF == 2, "register sets per slice parity");
constexpr int IS_AD = MG_IS_AD;          // A-fragment ring: row tiles in flight + the one being multiplied (3, 4: no faster, NOTES_r06 6)

__global__ __launch_bounds__(IS_THR) void imgbank_split_kernel(const float* __restrict__ feat, int K, int P,
                                                              const uint4* __restrict__ Wh, const uint4* __restrict__ Wl,
                                                              const float* __restrict__ bias, int N,
                                                              float* __restrict__ bank, float* __restrict__ pooled,
                                                              unsigned short* __restrict__ bank_hi,
                                                              unsigned short* __restrict__ bank_lo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* Ahi = reinterpret_cast<uint4*>(smem);                                   // [2][IS_ROWS][IS_STR]
    uint4* Alo = Ahi + 2 * IS_ROWS * IS_STR;
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

    // ---- producer side (all waves): wave = chunk (8 feature rows of the slice = one 16-B chunk of 8 consecutive k per region row);
    //      the two 32-lane HALVES of the wave take four of its rows each, lane & 31 = region quad.  (Rounds 3-5: lane = region quad and
    //      all eight rows in the lane -- 28 of 64 lanes converting 32 elements each; now 56 lanes convert 16 each and the halves swap
    //      their packed half chunks (v_permlane32_swap) so that every lane still writes whole 16-B chunks: the lower half the region
    //      rows 4q, 4q + 1 of its quad, the upper half 4q + 2, 4q + 3.)
    const int hw = lane >> 5, ql = lane & 31;
    const int pq = ql < nq ? ql : nq - 1;                       // idle lanes repeat the last quad (unconditional loads), never store
    const bool st_on = ql < nq;
    // feature rows in flight: IS_PF slices ahead of their conversion (IS_PF register sets of four rows; slice c lives in set c % IS_PF)
    f32x4 slr[IS_PF][4];
    auto gload = [&](int c, auto setc) {
        constexpr int set = decltype(setc)::v;
        const int cc = c < nk ? c : nk - 1;                      // (unconditional: behind the last slice the last one again)
#pragma unroll
        for (int i = 0; i < 4; ++i) slr[set][i] = *reinterpret_cast<const f32x4*>(fb + (size_t)(cc * IS_BK + wave * 8 + hw * 4 + i) * P + 4 * pq);
    };
    // cs: the slice in the registers (its pooled maxima are stored from here; < 0: none -- the extra conversion behind the last slice)
    auto emit = [&](int buf, int cs, auto setc) {
        f32x4 (&sl)[4] = slr[decltype(setc)::v];
        // pooled maxima of the slice's rows over this half's regions, without LDS: the lane's maximum over its four regions, then over
        // the 16-lane rows by DPP (quad_perm x 2, row_half_mirror, row_mirror), then row_bcast:15 folds the first row of each 32-lane
        // half into the second -- lanes 16-31 / 48-63 hold the half's maximum of feature row i; lanes 16-19 / 48-51 store the four rows.
        // (Rounds 3-5: partial maxima through an LDS tile and a second pass of all 512 threads at the head of the next slice: 590
        // cycles per slice and wave.)  One asm block, the four rows interleaved: a DPP read of a VGPR needs two wait states behind
        // the VALU write (three other instructions in between here; s_nop 1 in front of the first).
        if (pooled) {
            float m0, m1, m2, m3;
            {
                const float ninf = -INFINITY;
                m0 = st_on ? fmaxf(fmaxf(sl[0][0], sl[0][1]), fmaxf(sl[0][2], sl[0][3])) : ninf;
                m1 = st_on ? fmaxf(fmaxf(sl[1][0], sl[1][1]), fmaxf(sl[1][2], sl[1][3])) : ninf;
                m2 = st_on ? fmaxf(fmaxf(sl[2][0], sl[2][1]), fmaxf(sl[2][2], sl[2][3])) : ninf;
                m3 = st_on ? fmaxf(fmaxf(sl[3][0], sl[3][1]), fmaxf(sl[3][2], sl[3][3])) : ninf;
            }
#define IS_MAX4(ctrl)                                                                   \
    "v_max_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                  \
    "v_max_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                  \
    "v_max_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                  \
    "v_max_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf\n\t"
            asm volatile("s_nop 1\n\t" IS_MAX4("quad_perm:[1,0,3,2]") IS_MAX4("quad_perm:[2,3,0,1]") IS_MAX4("row_half_mirror") IS_MAX4("row_mirror")
                         "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                         "v_max_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                         "v_max_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                         "v_max_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                         "s_nop 1"
                         : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3));
#undef IS_MAX4
            const int r4 = lane & 3;
            const float mv = r4 == 0 ? m0 : r4 == 1 ? m1 : r4 == 2 ? m2 : m3;
            if (cs >= 0 && (lane & 0x1C) == 0x10) pooled[((size_t)b * 2 + mh) * K + cs * IS_BK + wave * 8 + hw * 4 + r4] = mv;
        }
        // this lane's four k (its half of the chunk) of the quad's four region rows: hi / lo images, two packed dwords each
        unsigned ch[4][2], cl[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float h[4], l[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float x = sl[i][j];
                h[i] = __builtin_bit_cast(float, (is_pack2(x, 0.f) << 16));              // bf16(x) as fp32
                l[i] = x - h[i];
            }
            ch[j][0] = is_pack2(h[0], h[1]); ch[j][1] = is_pack2(h[2], h[3]);
            cl[j][0] = is_pack2(l[0], l[1]); cl[j][1] = is_pack2(l[2], l[3]);
        }
        // v_permlane32_swap a, b: a = [a.lo, b.lo], b = [a.hi, b.hi].  With a = region row j, b = region row j + 2: the lower half
        // ends with (its own k 0-3, the upper half's k 4-7) of row j, the upper half with (the lower half's k 0-3, its own k 4-7) of
        // row j + 2 -- a whole chunk per lane and row pair.  (All 64 lanes take part; s_nop: the VALU write -> swap read and swap
        // write -> VALU read hazards the compiler does not see through an asm block.)
        asm volatile("s_nop 1\n\t"
                     "v_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\tv_permlane32_swap_b32 %3, %7\n\t"
                     "v_permlane32_swap_b32 %8, %12\n\tv_permlane32_swap_b32 %9, %13\n\tv_permlane32_swap_b32 %10, %14\n\tv_permlane32_swap_b32 %11, %15\n\t"
                     "s_nop 1"
                     : "+v"(ch[0][0]), "+v"(ch[0][1]), "+v"(ch[1][0]), "+v"(ch[1][1]), "+v"(ch[2][0]), "+v"(ch[2][1]), "+v"(ch[3][0]), "+v"(ch[3][1]),
                       "+v"(cl[0][0]), "+v"(cl[0][1]), "+v"(cl[1][0]), "+v"(cl[1][1]), "+v"(cl[2][0]), "+v"(cl[2][1]), "+v"(cl[3][0]), "+v"(cl[3][1]));
        if (st_on) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = 4 * ql + 2 * hw + r;
                const int at = (buf * IS_ROWS + row) * IS_STR + (wave ^ ((row >> 4) & 7));
                Ahi[at] = uint4{ch[r][0], ch[r][1], ch[r + 2][0], ch[r + 2][1]};
                Alo[at] = uint4{cl[r][0], cl[r][1], cl[r + 2][0], cl[r + 2][1]};
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
    const bool has3 = wave + 16 < IS_NT && (wave + 16) * 16 < N;      // (wave-uniform: a scalar branch)
    uint4 bh[IS_WD][IS_TPW], bl[IS_WD][IS_TPW];
    auto wload = [&](int ks, int slot) {
#pragma unroll
        for (int t = 0; t < IS_TPW; ++t) {
            bh[slot][t] = Wh[woff[t] + (size_t)ks * 64];
            bl[slot][t] = Wl[woff[t] + (size_t)ks * 64];
        }
    };
    const int fr = lane & 15, fg = lane >> 4;

    gload(0, mg_mha::IC<0>{});
#pragma unroll
    for (int d = 0; d < IS_WD - 1; ++d)
        if (d < KS) wload(d, d);
    if constexpr (IS_PF == 2) gload(1, mg_mha::IC<1 % IS_PF>{});
    emit(0, 0, mg_mha::IC<0>{});
    if constexpr (IS_PF == 2) gload(2, mg_mha::IC<0>{});
    else gload(1, mg_mha::IC<0>{});
    __syncthreads();
    // The two waves of a SIMD (w and w + 4) run a slice's two phases in OPPOSITE order: waves 0-3 first convert the next slice into
    // the other LDS buffer (VALU + LDS writes), then multiply; waves 4-7 multiply first.  While one wave of the SIMD holds the
    // matrix pipe the other converts -- in lock step (round 4) the pipe idled through every conversion and both waves then
    // contended for it.  The feature rows of slice c + 2 are requested right behind the conversion of slice c + 1 (the same 32
    // registers), i.e. at least one multiply phase before their use.
    // Every global load of the loop is UNCONDITIONAL (indices clamped to the last slice / k-step) and the two orders are two
    // instances of the loop, not a branch inside it: the compiler counts the loads in flight per path, and behind a join of paths
    // with different counts it waits for the weight fragments with the smaller count -- i.e. for the feature rows requested after
    // them, an HBM round trip in every k-step.
    // one slice; PH = c % 6 is a compile-time constant (the loop below is unrolled by six), so the LDS buffer c & 1 and the ring slots
    // (2 c + s) % 3 of its two k-steps are static register indices
    // (measured, NOTES_r05: the opposite orders buy little -- 233 us against 234 in lock step -- because the slice is bound by what ONE
    //  wave does in a row: conversion 1570 + pooled maxima 590 + 126 MFMAs 2680 cycles of a 6100-cycle slice for a three-tile wave.
    //  All loads as inline asm with hand-counted waits -- the compiler waits for a fragment with the count of the path that skipped
    //  both `if (convert_first)` regions, vmcnt(12) where 20 loads follow -- changed nothing either: 230 us.)
    const bool convert_first = wave < 4;
#ifdef MG_IS_TRACE
    unsigned long long tt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#endif
    auto slice = [&](auto phc, int c) {
        constexpr int PH = decltype(phc)::v;
        constexpr int buf = PH & 1;
        auto convert = [&]() {
            constexpr int set = (PH + 1) % IS_PF;               // slice c + 1's registers (PH = c % 6, IS_PF divides 6)
            emit(buf ^ 1, c + 1 < nk ? c + 1 : -1, mg_mha::IC<set>{});      // (behind the last slice: into the idle buffer, never read)
            gload(c + 1 + IS_PF, mg_mha::IC<set>{});
        };
        if (convert_first) convert();
        IS_T(0)                                                 // conversion at the head of the slice (waves 0-3)
        IS_T(1)                                                 // (rounds 3-5: the pooled maxima pass; now inside the conversion)
        mg_mha::static_for<0, 2>([&](auto sc) {
            constexpr int s = decltype(sc)::v;
            constexpr int SL = (2 * PH + s) % IS_WD;
            const int ks = c * 2 + s;
            wload(ks + IS_WD - 1 < KS ? ks + IS_WD - 1 : KS - 1, (SL + IS_WD - 1) % IS_WD);
            IS_T(2)                                             // fragment requests
            // Two branch-free blocks per k-step (a branch per (row tile, column tile) cut the MFMA stream into blocks of three, each
            // behind its own LDS wait): every wave's first two column tiles for ALL seven row tiles (row tiles behind the half's
            // last region multiply rows nobody stores), then -- waves 0-2 only, one scalar branch -- the third column tile with the
            // A fragments read a second time.
            // (A fragments one row tile ahead, by hand, with a scheduling fence per row tile: left alone the scheduler hoists all
            //  fourteen reads of a block in front of its MFMAs -- 127 spilled registers)
            // (round 6, measured: a ring of three or four -- two / three row tiles ahead, MG_IS_AD -- changes nothing: 208.7-219.9 us whatever
            //  the depth, three alternating rounds; the fragment reads are not what the MFMA blocks wait for)
            uint4 ah[IS_AD], al[IS_AD];
            auto aread = [&](int i, int slot) {
                const int at = (buf * IS_ROWS + i * 16 + fr) * IS_STR + ((4 * s + fg) ^ (i & 7));
                ah[slot] = Ahi[at];
                al[slot] = Alo[at];
            };
#pragma unroll
            for (int d = 0; d < IS_AD - 1; ++d) aread(d, d);
            mg_mha::static_for<0, IS_MT>([&](auto ic) {
                constexpr int i = decltype(ic)::v;
                if constexpr (i + IS_AD - 1 < IS_MT) aread(i + IS_AD - 1, (i + IS_AD - 1) % IS_AD);
                __builtin_amdgcn_sched_barrier(0);              // (the reads stay IN FRONT of this tile's MFMAs: they were sunk behind four of them)
                const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[i % IS_AD]), xl = __builtin_bit_cast(bf16x8, al[i % IS_AD]);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bf16x8 wh = __builtin_bit_cast(bf16x8, bh[SL][t]), wl = __builtin_bit_cast(bf16x8, bl[SL][t]);
                    acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wl, acc[i][t], 0, 0, 0);
                    acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, wh, acc[i][t], 0, 0, 0);
                    acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wh, acc[i][t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if (has3) {
                // the third column tile (waves 0-2), row tiles in PAIRS: three products on ONE accumulator in a row wait for each other
                // (32 instead of 16 cycles per MFMA: rounds 3-5 ran this pass tile by tile); two row tiles give two independent chains.
                // (Both passes as one -- two or three column tiles per row tile behind a wave-uniform branch -- is two instances of the
                //  block: 395 spilled registers.)
                const bf16x8 wh = __builtin_bit_cast(bf16x8, bh[SL][2]), wl = __builtin_bit_cast(bf16x8, bl[SL][2]);
                uint4 ph[4], pl[4];
                auto pread = [&](int i, int slot) {
                    const int at = (buf * IS_ROWS + i * 16 + fr) * IS_STR + ((4 * s + fg) ^ (i & 7));
                    ph[slot] = Ahi[at];
                    pl[slot] = Alo[at];
                };
                pread(0, 0);
                pread(1, 1);
                mg_mha::static_for<0, (IS_MT + 1) / 2>([&](auto pc) {
                    constexpr int i0 = 2 * decltype(pc)::v, i1 = i0 + 1;
                    if constexpr (i0 + 2 < IS_MT) pread(i0 + 2, (i0 + 2) & 3);
                    if constexpr (i0 + 3 < IS_MT) pread(i0 + 3, (i0 + 3) & 3);
                    __builtin_amdgcn_sched_barrier(0);
                    const bf16x8 xh0 = __builtin_bit_cast(bf16x8, ph[i0 & 3]), xl0 = __builtin_bit_cast(bf16x8, pl[i0 & 3]);
                    if constexpr (i1 < IS_MT) {
                        const bf16x8 xh1 = __builtin_bit_cast(bf16x8, ph[i1 & 3]), xl1 = __builtin_bit_cast(bf16x8, pl[i1 & 3]);
                        acc[i0][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh0, wl, acc[i0][2], 0, 0, 0);
                        acc[i1][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh1, wl, acc[i1][2], 0, 0, 0);
                        acc[i0][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl0, wh, acc[i0][2], 0, 0, 0);
                        acc[i1][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl1, wh, acc[i1][2], 0, 0, 0);
                        acc[i0][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh0, wh, acc[i0][2], 0, 0, 0);
                        acc[i1][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh1, wh, acc[i1][2], 0, 0, 0);
                    } else {
                        acc[i0][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh0, wl, acc[i0][2], 0, 0, 0);
                        acc[i0][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl0, wh, acc[i0][2], 0, 0, 0);
                        acc[i0][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh0, wh, acc[i0][2], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            IS_T(3)                                             // the k-step's MFMA blocks (incl. the wait for its fragments)
        });
        if (!convert_first) convert();
        IS_T(4)                                                 // conversion at the tail (waves 4-7)
        __syncthreads();
        IS_T(5)                                                 // barrier
    };
    static_assert(IS_WD == 3, "the slice loop is unrolled for a ring of three");
    for (int c = 0; c < nk; c += 6) {
        slice(mg_mha::IC<0>{}, c);
        if (c + 1 < nk) slice(mg_mha::IC<1>{}, c + 1);
        if (c + 2 < nk) slice(mg_mha::IC<2>{}, c + 2);
        if (c + 3 < nk) slice(mg_mha::IC<3>{}, c + 3);
        if (c + 4 < nk) slice(mg_mha::IC<4>{}, c + 4);
        if (c + 5 < nk) slice(mg_mha::IC<5>{}, c + 5);
    }
#ifdef MG_IS_TRACE
    if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4))
        for (int i = 0; i < 8; ++i) g_is_trace[wave >> 2][i] = tt[i];
#endif
    auto epilogue = [&]() {
    // ---- epilogue: + bias; acc[i][t][r] = bank[p0 + 16 i + 4 (lane >> 4) + r][16 nt + (lane & 15)] ----
    if (bank) {
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
    if (bank_hi) {
        // split-bf16 images: a lane and its right neighbour (quad_perm [1,1,3,3]: pure VALU) give one packed pair of columns, the
        // even lanes store 4 bytes to each image; columns [N, 320) are zero (N even: the launcher checks)
        unsigned* oh = reinterpret_cast<unsigned*>(bank_hi + ((size_t)b * P + p0) * IS_LD);
        unsigned* ol = reinterpret_cast<unsigned*>(bank_lo + ((size_t)b * P + p0) * IS_LD);
#pragma unroll
        for (int t = 0; t < IS_TPW; ++t) {
            const int n = (wave + 8 * t) * 16 + fr;
            const bool on = ton[t] && n < N;
            const float bv = (on && bias) ? bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < IS_MT; ++i) {
                if (i >= mtn) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int p = i * 16 + fg * 4 + r;
                    const float v = acc[i][t][r] + bv;
                    const float vn = MG_DPP(v, 0xF5);
                    const unsigned h2 = is_pack2(v, vn);
                    const unsigned l2 = is_pack2(v - __builtin_bit_cast(float, h2 << 16), vn - __builtin_bit_cast(float, h2 & 0xFFFF0000u));
                    if (on && !(fr & 1) && p < rows) {
                        oh[((size_t)p * IS_LD + n) >> 1] = h2;
                        ol[((size_t)p * IS_LD + n) >> 1] = l2;
                    }
                }
            }
        }
        const int zc = (IS_LD - N) >> 1;                               // zero pairs per row
        for (int idx = tid; idx < rows * zc; idx += IS_THR) {
            const int p = idx / zc, q = idx - p * zc;
            oh[((size_t)p * IS_LD + N) / 2 + q] = 0u;
            ol[((size_t)p * IS_LD + N) / 2 + q] = 0u;
        }
    }
    };
    epilogue();
}

}  // namespace

#ifdef MG_IS_TRACE
extern "C" int mgnns_debug_is_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_is_trace), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int mgnns_imgbank_pool_split_fwd(const float* feat, int B, int K, int P, const void* Wp_hi, const void* Wp_lo,
                                            const float* bias, int N, float* bank, float* pooled_halves, void* bank_hi,
                                            void* bank_lo, mgnns_stream_t stream) {
    MG_REQUIRE(feat && Wp_hi && Wp_lo && (bank || bank_hi), "mgnns_imgbank_pool_split_fwd: null pointer");
    MG_REQUIRE(!bank_hi == !bank_lo, "mgnns_imgbank_pool_split_fwd: bank_hi and bank_lo come together");
    MG_REQUIRE(!bank_hi || (N % 2 == 0 && mg_aligned16(bank_hi) && mg_aligned16(bank_lo)),
               "mgnns_imgbank_pool_split_fwd: the split-bf16 images need an even N (%d) and 16-byte aligned buffers", N);
    MG_REQUIRE(B >= 0 && K > 0 && K % IS_BK == 0, "mgnns_imgbank_pool_split_fwd: K=%d must be a positive multiple of %d", K, IS_BK);
    MG_REQUIRE(P > 0 && P % 4 == 0 && P <= IS_PSPLIT + IS_ROWS, "mgnns_imgbank_pool_split_fwd: P=%d unsupported (multiple of 4, <= %d)", P,
               IS_PSPLIT + IS_ROWS);
    MG_REQUIRE(N > 0 && N <= IS_NT * 16, "mgnns_imgbank_pool_split_fwd: N=%d unsupported (<= %d)", N, IS_NT * 16);
    MG_REQUIRE(mg_aligned16(feat) && mg_aligned16(Wp_hi) && mg_aligned16(Wp_lo), "mgnns_imgbank_pool_split_fwd: feat / weights must be 16-byte aligned");
    if (B == 0) return 0;
    const size_t lds = (size_t)4 * IS_ROWS * IS_STR * 16;
    MG_DYN_LDS(imgbank_split_kernel, lds);
    hipLaunchKernelGGL(imgbank_split_kernel, dim3(2 * B), dim3(IS_THR), lds, (hipStream_t)stream, feat, K, P,
                       reinterpret_cast<const uint4*>(Wp_hi), reinterpret_cast<const uint4*>(Wp_lo), bias, N, bank, pooled_halves,
                       reinterpret_cast<unsigned short*>(bank_hi), reinterpret_cast<unsigned short*>(bank_lo));
    MG_CHECK_LAUNCH("mgnns_imgbank_pool_split_fwd");
    return 0;
}
