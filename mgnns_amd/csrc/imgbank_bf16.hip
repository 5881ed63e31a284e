// bf16-operand image memory bank + global max-pool, one pass over the fp32 [B,2048,196] feature map
// (get_img_*_memory_bank, MODEL:400-428, fused with MaxPool2d(14,14), MODEL:454-455):
//   bank[b,p,:] = bf16( W * feat[b,:,p] + bias )   -> [B, P, 320] bf16 (model dim zero padded: the layout the
//                                                     bf16 fusion-attention kernel stages straight into LDS)
//   pooled[b,k] = max_p feat[b,k,p]                 (exact fp32)
// The kernel is HBM bound on the fp32 map (1.6 MB per sample-channel, read exactly once).
//
// Round 3 design: ONE 512-thread workgroup per sample; the map never touches a VGPR on its way in.  The sample is a
// contiguous 1.6-MB block whose rows (one feature k, all P regions: 784 B) arrive by LDS-DMA (buffer_load_dwordx4 ... lds) into a
// ring of five k-step slots (32 rows = 24.5 KB each): ~75 KB in flight per CU without a register or a producer wave tied up,
// which is what the stream needs -- the same ring with a trivial consumer moves 6.6 TB/s (tools/dev/micro/dma_stream.hip),
// against 3.5 TB/s for rounds 1-2's register path (two workgroups per sample, each reading 448-B half rows: 1.19x sector
// over-fetch; producer waves whose every computing cycle was a cycle the memory pipe was not refilled).
// Per k-step all eight waves (a) turn the landed fp32 slot into the bf16 MFMA operand image [p][32 k] (LDS -> registers ->
// v_cvt_pk_bf16_f32 -> LDS, 16-B chunks swizzled so that both sides are conflict-free), (b) take the max-pool of its 32 rows,
// (c) after ONE barrier run the k-step's MFMAs: wave (rg, cg) owns row tiles 7 rg .. and column tiles 5 cg .. (7 x 5 tiles,
// v_mfma_f32_16x16x32_bf16, W fragments fragment-major from L2, two k-steps ahead).  Epilogue through LDS: bank rows leave
// as 16-B lanes.
#include "common.hpp"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifdef MG_IMG_TRACE
// profiling aid (off by default): per-phase cycle sums of wave 0 / wave 7 of two workgroups
__device__ unsigned long long g_img_trace[4][8];
#define IMG_T() __builtin_amdgcn_s_memtime()
#endif

#ifndef MG_IMG_AUX
#define MG_IMG_AUX 2          // cache policy of the map loads (aux bits of buffer_load): 2 = non-temporal
#endif

namespace {

constexpr int BK = 64;                  // K must be a multiple of this (two k-steps per trip of the main loop)
constexpr int MT = 13;                  // row tiles of 16 regions (P <= 208)
constexpr int NT = 19;                  // 304 / 16
constexpr int OUT_LD = 320;             // bank row length (bf16)
constexpr int OCH = OUT_LD / 8;         // 40 chunks per output row
constexpr int OSTR = OUT_LD * 2 + 16;   // epilogue LDS row stride in bytes (656: rows land on distinct banks)
constexpr int NTHR = 512;

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

// ---- LDS map ------------------------------------------------------------------------------------------------------------
constexpr int KROWS = 32;                                // feature rows per k-step
constexpr int UROWS = 4;                                 // ... per ring unit: HALF a k-octet, one wave's share of a k-step
constexpr int NUS = 3;                                   // unit slots of a wave's private ring
constexpr int PMAX = MT * 16;                            // 208 regions at most
constexpr int USLOT = UROWS * PMAX * 4;                  // 3,328 B: a unit holds 4 rows of P * 4 bytes, packed
constexpr int WRING = NUS * USLOT;                       // 9,984 B per wave
constexpr int ABUF = PMAX * 64;                          // 13,312 B: bf16 operand image of one k-step, [p][4 chunks of 8 k]
constexpr int WBUF = (NT + 1) * 1024;                    // 20 KB: the 19 W fragments of one k-step (+ 1 KB that takes a dummy request)
constexpr size_t OFF_A = (size_t)8 * WRING;
constexpr size_t OFF_W = OFF_A + 2 * ABUF;
constexpr size_t SMEM_BYTES = OFF_W + 2 * WBUF;
static_assert(SMEM_BYTES <= 160 * 1024, "LDS");
static_assert((size_t)PMAX * OSTR <= SMEM_BYTES, "the epilogue stages the bank rows over the ring");
#ifndef MG_IMG_NRW
#define MG_IMG_NRW 6
#endif
constexpr int NRW = MG_IMG_NRW, NRM = MT - NRW;          // row tiles of a W wave (it also requests the W fragments) / of the others

// LDS accesses of the main loop are inline asm: hipcc puts s_waitcnt vmcnt(0) in front of every LDS access it can see while an
// LDS-DMA may be in flight, which would drain the ring at every step.  The caller orders them (lgkmcnt / barriers).
template <int N> struct ICI { static constexpr int v = N; };
template <int I, int N, typename F>
__device__ __forceinline__ void static_for_img(F&& f) {
    if constexpr (I < N) {
        f(ICI<I>{});
        static_for_img<I + 1, N>(f);
    }
}

// (v_max_f32 through asm: fmaxf costs a canonicalising v_max_f32 v, v, v per operand in front of the real one)
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// Max over the four 16-lane rows of the wave for FOUR values at once (a transposing reduction, sq_mha_bf16.hip::rows4_sum4 with
// max): on return the rows of the result hold [a, c, b, d] folded over the rows.
__device__ __forceinline__ float rows4_max4(float a, float b, float c, float d) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    float ab = vmax(a, b), cd = vmax(c, d);
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(ab), "+v"(cd));
    return vmax(ab, cd);
}
// max over each 16-lane row, the DPP operand folded into the v_max_f32 (one instruction per level; s_nop 1 = the two wait states
// between a VALU write and a DPP read of the same register, which the compiler does not see through an asm block)
__device__ __forceinline__ float row16_max_dpp(float a) {
    asm volatile("s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                 : "+v"(a));
    return a;
}

// One wave's share of a sample.  Round 4: every wave owns HALF a k-octet of every k-step end to end -- wave w = (octet w & 3,
// half w >> 2) requests those four rows (LDS-DMA into its private ring of three units), waits for them on its OWN vmcnt, turns
// them into its 8-byte halves of the octet's chunks of the bf16 operand image, takes their max-pool from the same registers and
// re-requests into the unit slot it has just read; all of it in the gaps of its MFMA stream.  Waves 0-3 ("W waves", WS) also
// request the W fragments and take NRW of the 13 row tiles, the others NRM.  Before (round 3): waves 4-7 requested all rows, waves
// 0-3 converted them, everybody pooled them from LDS a second time -- "landed" had to be published through the step's barrier a
// step early, the pool's reads sat inside the MFMA stream's counted waits, the compute on top of the stream was additive
// (ablations: profiles/NOTES_r04.md section 7).  Vector-memory operations of a wave complete IN ORDER: a W wave requests its
// fragments of the next k-step BEFORE the step's rows, so that they wait only for rows requested a step earlier.
template <int NR, int NW, bool WS, int PT>
__device__ __forceinline__ void img_wave(unsigned char* smem, const float* __restrict__ feat, int b, int K, int Prt,
                                         const unsigned short* __restrict__ Wp, const float* __restrict__ bias, int N,
                                         float* __restrict__ pooled_part, float* __restrict__ pooled, int wave, int lane) {
    const int tile0 = WS ? 0 : NRW, cg = wave & 3, oct = wave & 3, half = wave >> 2;
    const int nks = K / KROWS;
    const int P = PT ? PT : Prt;                         // PT: the region count as a compile-time constant (row offsets become immediates)
    const int RB = P * 4;                                // bytes of a map row
    const unsigned lds0 = mg_lds_addr(smem), abuf = lds0 + (unsigned)OFF_A, wbuf = lds0 + (unsigned)OFF_W;
    const unsigned myring = lds0 + (unsigned)(wave * WRING);
    const __amdgpu_buffer_rsrc_t f_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(feat + (size_t)b * K * P), 0, K * P * (int)sizeof(float), 0x00027000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(Wp), 0, NT * nks * 1024, 0x00027000);
    const int nquad = P >> 2;

    // ---- the map stream: unit j of this wave = rows 8 oct + 4 half .. + 3 of k-step j -> unit slot j % 3; one row (P / 4 lanes
    //      x 16 B) per DMA instruction; units past the end are out of range and arrive as zeros in a slot nobody reads (the
    //      request count per step stays constant: the counted waits depend on it)
    const bool dma_lane = lane < nquad;
    auto dma_rows = [&](int j, int r0, int nr) {
        if (dma_lane) {
            const int row0 = j * KROWS + oct * 8 + half * UROWS;
#pragma unroll
            for (int r = r0; r < r0 + nr; ++r) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    f_rsrc, (__attribute__((address_space(3))) void*)(uintptr_t)(smem + (size_t)wave * WRING + (size_t)(j % NUS) * USLOT + r * RB),
                    16, lane * 16, (row0 + r) * RB, 0, MG_IMG_AUX);
            }
        }
    };
    // ---- the W stream (W waves): the 19 fragments (1 KiB each, fragment-major in memory) of k-step s -> W image s & 1; wave w
    //      takes fragments w, w + 4, ..: five requests each, the 20th is a dummy (out of range: zeros into the image's spare KB)
    auto dma_w = [&](int s, int j) {
        const int t = wave + 4 * j;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            w_rsrc, (__attribute__((address_space(3))) void*)(uintptr_t)(smem + OFF_W + (size_t)(s & 1) * WBUF + t * 1024), 16,
            lane * 16, t < NT && s < nks ? (t * nks + s) * 1024 : 0x7ffffc00, 0, 0);
    };

    f32x4 acc[NR][NW];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int t = 0; t < NW; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // operand image: [p][4 chunks of 16 B], chunk c of row p at slot c ^ g[(p >> 2) & 3], g = {2, 0, 1, 3}: the MFMA operand
    // reads are conflict-free (a ds_read_b128's 16-lane groups take rows 0-3 / 12-15 of chunk c and rows 4-11 of chunk c ^ 1:
    // slots {c^2, c^3} and {c^1^0, c^1^1} = all four, each with the four rows of a 256-B bank line) and the conversion's
    // 8-byte writes (16 consecutive lanes = 16 consecutive region quads, rows 256 B apart) spread over all four slots
    auto a_off = [](int p, int c) { return (unsigned)(p * 64 + (((c ^ (0xD2 >> (2 * ((p >> 2) & 3)))) & 3) << 4)); };
    // conversion task of a lane: region quad pq = lane (lanes past the last quad read the last quad again: their maxima change
    // nothing, their image writes go to the W image's spare KB), the wave's half octet: 4 rows x 4 regions -> four 8-B half chunks
    // + 4 row maxima
    const int pq = lane < nquad ? lane : nquad - 1;
    const unsigned cv_dst = lane < nquad ? abuf + a_off(4 * pq, oct) + (unsigned)(half * 8)      // rows 4 pq .. + 3 share (p >> 2): same slot
                                         : wbuf + (unsigned)(NT * 1024 + (lane & 15) * 16);
    const unsigned cv_step = lane < nquad ? (unsigned)ABUF : 0u;                                 // image s1 & 1
    // MFMA operand reads: map fragment = row 16 (tile0 + i) + (lane & 15), chunk lane >> 4; W fragment t of this wave
    const unsigned a_rd = abuf + (unsigned)(tile0 * 16 * 64) + a_off(lane & 15, lane >> 4);
    const unsigned w_rd = wbuf + (unsigned)(cg * 5 * 1024 + lane * 16);

    // pooled maxima leave as ONE store per wave and k-step: after the transposing reduction the 16-lane rows of the wave hold the
    // maxima of the unit's rows {0, 2, 1, 3}; lanes 0 .. of a row store it to the two partial arrays (both get the complete
    // maxima: callers that combine them keep working) and, if asked for, the combined one
    const int grp = lane >> 4, sub = lane & 15;
    const bool pst_on = sub < (pooled ? 3 : 2);
    float* const pst = (sub == 0 ? pooled_part + (size_t)b * 2 * K : sub == 1 ? pooled_part + ((size_t)b * 2 + 1) * K : pooled + (size_t)b * K) +
                       oct * 8 + half * UROWS + ((grp & 1) << 1 | (grp >> 1));
    auto max3 = [](float a, float b2, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b2), "v"(c)); return r; };
    // ---- this wave's unit of k-step s1 (landed: see the counted wait at the call) -> its half of chunk `oct` of the bf16 operand
    //      image A[s1 & 1] + the max-pool of its four rows.  In PIECES (cv_piece<k>) for the gaps of the MFMA stream: reads |
    //      half chunk of region e (2 conversions + one 8-B write) x 4 | row maxima x 2 | the transposing reduction | the store
    f32x4 v[4];
    float m[4];
    unsigned cv_ab = 0;
    auto cv_reads = [&](int s1) {
        cv_ab = cv_dst + (s1 & 1) * cv_step;
        const unsigned ua = myring + (unsigned)((s1 % NUS) * USLOT + pq * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (PT > 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[j]) : "v"(ua), "n"(j * PT * 4) : "memory");
            else asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(ua + j * RB) : "memory");
        }
    };
    // (pieces of at most three instructions: a SIMD issues about that much in the shadow of one 16-cycle MFMA; the s_nop are the
    //  wait states of the lane-swap / DPP operands, which the compiler does not see through the asm blocks)
    float mab = 0.f, mcd = 0.f;
    auto cv_piece = [&](auto kc, int s1) {
        constexpr int k = decltype(kc)::v;
        if constexpr (k < 4) {                           // half chunk of region 4 pq + k
            u32x2 c;
            c[0] = pack2(v[0][k], v[1][k]); c[1] = pack2(v[2][k], v[3][k]);
            asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(cv_ab), "v"(c), "n"(k * 64) : "memory");
        } else if constexpr (k < 8) {                    // maximum of row k - 4 over this lane's four regions
            constexpr int j = k - 4;
            m[j] = vmax(max3(v[j][0], v[j][1], v[j][2]), v[j][3]);
        } else if constexpr (k == 8) {                   // the transposing reduction over the four 16-lane rows of the wave ...
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3" : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]));
        } else if constexpr (k == 9) {
            asm volatile("s_nop 1\n\tv_max_f32 %0, %2, %3\n\tv_max_f32 %1, %4, %5" : "=&v"(mab), "=&v"(mcd) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]));
        } else if constexpr (k == 10) {
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(mab), "+v"(mcd));
        } else if constexpr (k == 11) {                  // ... rows hold unit rows {0, 2, 1, 3}
            asm volatile("s_nop 1\n\tv_max_f32 %0, %1, %2" : "=v"(m[0]) : "v"(mab), "v"(mcd));
        } else if constexpr (k == 12) {                  // the 16 lanes of a row, DPP operand folded into the v_max_f32
            asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(m[0]));
        } else if constexpr (k == 13) {
            asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(m[0]));
        } else if constexpr (k == 14) {
            asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(m[0]));
        } else if constexpr (k == 15) {
            asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(m[0]));
        } else if constexpr (k == 16) {
            if (pst_on) pst[s1 * KROWS] = m[0];
        }
    };
    constexpr int NCV = 17;                              // conversion pieces
    constexpr int ND = NW >= 5 ? 4 : 2;                  // gaps the unit's four row requests are spread over
    constexpr int Q0 = (WS ? 5 : 0) + ND;                // MFMAs in front of the first conversion piece
    static_assert(Q0 + NCV <= NR * NW, "one piece per MFMA gap");
    // ---- the MFMAs of k-step s (map fragments from A[s & 1], W fragments from W[s & 1]), and IN THEIR GAPS -- a SIMD issues
    //      about one other instruction in the shadow of a 16-cycle MFMA -- behind MFMA number q of the step: a W wave's request q
    //      of the next W image (q < 5), then the four row requests of unit s + 3 (whose slot the conversion of the step before has
    //      read), then the conversion pieces of k-step s + 1, whose reads were issued in front of the operand reads (LDS
    //      operations complete in order)
    auto mfma_step = [&](int s, auto cvc) {
        constexpr bool cv = decltype(cvc)::v != 0;
        const unsigned ab = (unsigned)((s & 1) * ABUF), wb = (unsigned)((s & 1) * WBUF);
        u32x4 wf[NW], bf[NR];
        if (cv) cv_reads(s + 1);
#pragma unroll
        for (int t = 0; t < NW; ++t) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[t]) : "v"(w_rd + wb), "n"(t * 1024) : "memory");
#pragma unroll
        for (int i = 0; i < NR; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bf[i]) : "v"(a_rd + ab), "n"(i * 1024) : "memory");
        static_for_img<0, NR>([&](auto ic) {
            constexpr int i = decltype(ic)::v;
            // row tile i's MFMAs wait for the W fragments and map fragments 0 .. i only: younger are the other fragments and the
            // image writes of the conversion pieces issued so far (pieces 0-3 sit behind MFMAs Q0 .. Q0 + 3)
            constexpr int wr = !cv ? 0 : (i * NW - Q0 < 0 ? 0 : i * NW - Q0 > 4 ? 4 : i * NW - Q0);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR - 1 - i + wr) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 bv = __builtin_bit_cast(bf16x8, bf[i]);
            static_for_img<0, NW>([&](auto tc) {
                constexpr int t = decltype(tc)::v, q = i * NW + t;
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[t]), bv, acc[i][t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (WS && q < 5) dma_w(s + 1, q);
                else if constexpr (q < Q0) dma_rows(s + NUS, (q - (Q0 - ND)) * (UROWS / ND), UROWS / ND);
                else if constexpr (q - Q0 < NCV) { if (cv) cv_piece(ICI<q - Q0>{}, s + 1); }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };
    // "My requests up to x have landed" is a COUNT (in-order completion): at most as many operations outstanding as this wave
    // has issued after them.  A wave's order: prologue [W(0) x 5] [units 0, 1, 2] [store 0]; step j [W(j+1) x 5] [unit j+3]
    // [store j+1] (the bracketed W requests: W waves only).
    if (WS) {
#pragma unroll
        for (int j = 0; j < 5; ++j) dma_w(0, j);
    }
#pragma unroll
    for (int j = 0; j < NUS; ++j) dma_rows(j, 0, UROWS);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UROWS * (NUS - 1)) : "memory");           // W(0) and unit 0
    cv_reads(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    static_for_img<0, NCV>([&](auto kc) { cv_piece(kc, 0); });
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#ifdef MG_IMG_TRACE
    unsigned long long tt[6] = {0, 0, 0, 0, 0, 0};
#define MG_TT(i, t0) { const unsigned long long t1_ = IMG_T(); tt[i] += t1_ - t0; t0 = t1_; }
#else
#define MG_TT(i, t0)
#endif
    // Step s: the MFMAs of k-step s on every wave, with the step's requests and the conversion + pool of the wave's unit of
    // k-step s + 1 (into the other operand image) in the gaps; one barrier.
    auto step = [&](int s, auto cvc) {
#ifdef MG_IMG_TRACE
        unsigned long long t0 = IMG_T();
#endif
        // unit s + 1 of this wave (requested in step s - 2); younger: store s - 1, [W(s) x 5,] unit s + 2, store s
        // (s = 0: unit 2 and store 0)
        if (s == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UROWS + 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UROWS + 2 + (WS ? 5 : 0)) : "memory");
        MG_TT(3, t0);
        mfma_step(s, cvc);
        MG_TT(2, t0);
        if (WS) {
            // the fragments of k-step s + 1; younger: unit s + 3 and (not in the last step) store s + 1
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UROWS + (decltype(cvc)::v ? 1 : 0)) : "memory");
            MG_TT(1, t0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        MG_TT(4, t0);
    };
    for (int s = 0; s + 1 < nks; ++s) step(s, ICI<1>{});
    step(nks - 1, ICI<0>{});                             // (nothing left to convert)
#ifdef MG_IMG_TRACE
    if (lane == 0 && (wave == 0 || wave == 4) && (blockIdx.x == 0 || blockIdx.x == 129)) {
        unsigned long long* g = g_img_trace[(blockIdx.x ? 2 : 0) + (wave >> 2)];
        for (int i = 0; i < 5; ++i) g[i] = tt[i];
    }
#endif
    // ---- epilogue: + bias, bf16, through LDS (over the ring: every request has landed), 16-B row stores; columns N..319 zero.
    // The tiles were computed TRANSPOSED (A operand = W fragment, B operand = map fragment): this lane's accumulator element
    // [i][t][r] is output column (5 cg + t) * 16 + 4 * (lane >> 4) + r of region row 16 (tile0 + i) + (lane & 15), i.e. four
    // CONSECUTIVE bank columns per tile -> one 8-byte LDS write.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned char* osb = smem;
#pragma unroll
    for (int t = 0; t < NW; ++t) {
        const int n = (cg * 5 + t) * 16 + (lane >> 4) * 4;
        f32x4 bvv = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bvv[r] = (n + r < N) ? bias[n + r] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int row = (tile0 + i) * 16 + (lane & 15);
            uint2 o;
            o.x = pack2(n + 0 < N ? acc[i][t][0] + bvv[0] : 0.f, n + 1 < N ? acc[i][t][1] + bvv[1] : 0.f);
            o.y = pack2(n + 2 < N ? acc[i][t][2] + bvv[2] : 0.f, n + 3 < N ? acc[i][t][3] + bvv[3] : 0.f);
            *reinterpret_cast<uint2*>(osb + (size_t)row * OSTR + n * 2) = o;
        }
    }
}

template <int PT>
__global__ __launch_bounds__(NTHR) void imgbank_pool_bf16_kernel(const float* __restrict__ feat, int Bn, int K, int P,
                                                                 const unsigned short* __restrict__ Wp,
                                                                 const float* __restrict__ bias, int N,
                                                                 unsigned short* __restrict__ bank,
                                                                 float* __restrict__ pooled_part, float* __restrict__ pooled) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    if ((wave >> 2) == 0) {
        if ((wave & 3) == 3) img_wave<NRW, 4, true, PT>(smem, feat, b, K, P, Wp, bias, N, pooled_part, pooled, wave, lane);
        else img_wave<NRW, 5, true, PT>(smem, feat, b, K, P, Wp, bias, N, pooled_part, pooled, wave, lane);
    } else {
        if ((wave & 3) == 3) img_wave<NRM, 4, false, PT>(smem, feat, b, K, P, Wp, bias, N, pooled_part, pooled, wave, lane);
        else img_wave<NRM, 5, false, PT>(smem, feat, b, K, P, Wp, bias, N, pooled_part, pooled, wave, lane);
    }
    // zero the pad columns 304..319 (two chunks per row), then the rows leave as 16-B lanes (all eight waves)
    unsigned char* osb = smem;
    for (int q = tid; q < P * 2; q += NTHR)
        *reinterpret_cast<uint4*>(osb + (size_t)(q >> 1) * OSTR + (NT * 2 + (q & 1)) * 16) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    uint4* ob = reinterpret_cast<uint4*>(bank) + (size_t)b * P * OCH;
    for (int q = tid; q < P * OCH; q += NTHR) {
        const int row = q / OCH, ch = q - row * OCH;
        ob[(size_t)row * OCH + ch] = *reinterpret_cast<const uint4*>(osb + (size_t)row * OSTR + ch * 16);
    }
}

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

// imgbank_bf16_pairs.hip: two workgroups per sample (region halves), for batches that do not fill the chip
int mg_imgbank_pool_bf16_pairs(const float* feat, int B, int K, int P, const void* Wp, const float* bias, int N, void* bank_bf16, int ld,
                               float* pooled, float* pooled_work, mgnns_stream_t stream);

static int g_imgbank_form = -1;          // -1: not set (MGNNS_IMGBANK_FORM or 0)
extern "C" int mgnns_imgbank_set_form(int form) {
    MG_REQUIRE(form >= 0 && form <= 2, "mgnns_imgbank_set_form: form=%d (0 by batch, 1 stream, 2 pairs)", form);
    g_imgbank_form = form;
    return 0;
}

extern "C" int mgnns_imgbank_pool_bf16_fwd(const float* feat, int B, int K, int P, const void* Wp, const float* bias, int N,
                                           void* bank_bf16, int ld, float* pooled, float* pooled_work,
                                           mgnns_stream_t stream) {
    MG_REQUIRE(feat && Wp && bank_bf16 && pooled_work, "mgnns_imgbank_pool_bf16_fwd: null pointer");
    MG_REQUIRE(B >= 0 && K > 0 && K % BK == 0, "mgnns_imgbank_pool_bf16_fwd: K=%d must be a positive multiple of %d", K, BK);
    MG_REQUIRE(P % 4 == 0 && P >= 16 && P <= PMAX, "mgnns_imgbank_pool_bf16_fwd: P=%d unsupported (multiple of 4 in [16, %d])", P, PMAX);
    MG_REQUIRE((double)K * P * 4 < 2147483648.0, "mgnns_imgbank_pool_bf16_fwd: a sample's map must stay below 2 GiB");
    MG_REQUIRE(N > 0 && N <= NT * 16, "mgnns_imgbank_pool_bf16_fwd: N=%d unsupported (<= %d)", N, NT * 16);
    MG_REQUIRE(ld == OUT_LD, "mgnns_imgbank_pool_bf16_fwd: bank row length must be %d", OUT_LD);
    MG_REQUIRE(mg_aligned16(feat) && mg_aligned16(Wp) && mg_aligned16(bank_bf16),
               "mgnns_imgbank_pool_bf16_fwd: feat/Wp/bank must be 16-byte aligned");
    if (B == 0) return 0;
    // A workgroup of the stream kernel needs ~95 us for its sample whatever the batch; up to half a chip of samples the
    // two-workgroups-per-sample form (half the chain per workgroup) is the faster one.  MGNNS_IMGBANK_FORM=1 / 2 forces one.
    const int n_cu = mg_cu_count();
    if (n_cu <= 0) return MGNNS_ERR_LAUNCH;
    const int form = g_imgbank_form >= 0 ? g_imgbank_form : mg_env_int("MGNNS_IMGBANK_FORM", 0, 3);
    const bool pairs_ok = K % 128 == 0 && P > 104 && P <= 200;
    if (pairs_ok && (form == 2 || (form == 0 && 2 * B <= n_cu)))
        return mg_imgbank_pool_bf16_pairs(feat, B, K, P, Wp, bias, N, bank_bf16, ld, pooled, pooled_work, stream);
    MG_DYN_LDS(imgbank_pool_bf16_kernel<196>, SMEM_BYTES);
    MG_DYN_LDS(imgbank_pool_bf16_kernel<0>, SMEM_BYTES);
    // one workgroup per sample; the two halves of pooled_work get the same (complete) maxima -- callers that combine them
    // themselves (label_tail) keep working --, `pooled` (optional) is written directly
    if (P == 196)             // the model's 14 x 14 maps: row offsets as instruction immediates
        hipLaunchKernelGGL(imgbank_pool_bf16_kernel<196>, dim3(B), dim3(NTHR), SMEM_BYTES, (hipStream_t)stream, feat, B, K, P,
                           reinterpret_cast<const unsigned short*>(Wp), bias, N, reinterpret_cast<unsigned short*>(bank_bf16),
                           pooled_work, pooled);
    else
        hipLaunchKernelGGL(imgbank_pool_bf16_kernel<0>, dim3(B), dim3(NTHR), SMEM_BYTES, (hipStream_t)stream, feat, B, K, P,
                           reinterpret_cast<const unsigned short*>(Wp), bias, N, reinterpret_cast<unsigned short*>(bank_bf16),
                           pooled_work, pooled);
    MG_CHECK_LAUNCH("mgnns_imgbank_pool_bf16_fwd");
    return 0;
}
