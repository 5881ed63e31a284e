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
constexpr int HROWS = 16;                                // ... per ring slot (half a k-step)
constexpr int NHS = 7;                                   // slots of the fp32 ring
constexpr int PMAX = MT * 16;                            // 208 regions at most
constexpr int HSLOT = HROWS * PMAX * 4;                  // 13,312 B: a slot holds 16 rows of P * 4 bytes, packed
constexpr int ABUF = PMAX * 64;                          // 13,312 B: bf16 operand image of one k-step, [p][4 chunks of 8 k]
constexpr int WBUF = (NT + 1) * 1024;                    // 20 KB: the 19 W fragments of one k-step (+ 1 KB that takes a dummy request)
constexpr size_t OFF_A = (size_t)NHS * HSLOT;
constexpr size_t OFF_W = OFF_A + 2 * ABUF;
constexpr size_t SMEM_BYTES = OFF_W + 2 * WBUF;
static_assert(SMEM_BYTES <= 160 * 1024, "LDS");
static_assert((size_t)PMAX * OSTR <= SMEM_BYTES, "the epilogue stages the bank rows over the ring");

// LDS accesses of the main loop are inline asm: hipcc puts s_waitcnt vmcnt(0) in front of every LDS access it can see while an
// LDS-DMA may be in flight, which would drain the ring at every step.  The caller orders them (lgkmcnt / barriers).
__device__ __forceinline__ void wait_vm(int n) {         // s_waitcnt vmcnt(n) for a run-time n (the immediate is an instruction field)
    switch (n) {
#define MG_VM_CASE(x) case x: asm volatile("s_waitcnt vmcnt(" #x ")" ::: "memory"); break;
        MG_VM_CASE(1) MG_VM_CASE(2) MG_VM_CASE(3) MG_VM_CASE(4) MG_VM_CASE(5) MG_VM_CASE(6) MG_VM_CASE(7) MG_VM_CASE(8)
        MG_VM_CASE(12) MG_VM_CASE(13) MG_VM_CASE(14) MG_VM_CASE(15) MG_VM_CASE(16) MG_VM_CASE(17) MG_VM_CASE(18) MG_VM_CASE(19)
        MG_VM_CASE(20) MG_VM_CASE(21) MG_VM_CASE(22) MG_VM_CASE(23) MG_VM_CASE(24)
#undef MG_VM_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <int N> struct ICI { static constexpr int v = N; };
template <int I, int N, typename F>
__device__ __forceinline__ void static_for_img(F&& f) {
    if constexpr (I < N) {
        f(ICI<I>{});
        static_for_img<I + 1, N>(f);
    }
}

// One wave's share of a sample.  NR x NW accumulator tiles: row tiles 7 rg .. 7 rg + NR - 1, column tiles 5 cg .. 5 cg + NW - 1.
// STAGER (row group 1 = waves 4-7): the wave requests map rows; otherwise (waves 0-3) the W fragments.  Vector-memory
// operations of a wave complete IN ORDER: a wave that has map rows in flight (HBM latency, several k-steps ahead) would get
// nothing else back before them -- W fragments requested into registers behind the rows made every k-step wait out the
// latency of rows it needs four steps later (122 us).  So the two streams are issued by DIFFERENT waves, both into LDS, and
// each wave's counted wait covers only its own stream.
template <int NR, int NW, bool STAGER, int PT>
__device__ __forceinline__ void img_wave(unsigned char* smem, const float* __restrict__ feat, int b, int K, int Prt,
                                         const unsigned short* __restrict__ Wp, const float* __restrict__ bias, int N,
                                         unsigned short* __restrict__ bank, float* __restrict__ pooled_part,
                                         float* __restrict__ pooled, int wave, int lane, int tid) {
    // row group: the W waves also convert, so they take the 6-tile group and the map waves the 7-tile one
    const int rg = STAGER ? 0 : 1, cg = wave & 3;
    const int nks = K / KROWS;
    const int P = PT ? PT : Prt;                         // PT: the region count as a compile-time constant (row offsets become immediates)
    const int RB = P * 4;                                // bytes of a map row
    const unsigned ring = mg_lds_addr(smem), abuf = ring + (unsigned)OFF_A, wbuf = ring + (unsigned)OFF_W;
    const __amdgpu_buffer_rsrc_t f_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(feat + (size_t)b * K * P), 0, K * P * (int)sizeof(float), 0x00027000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(Wp), 0, NT * nks * 1024, 0x00027000);

    // ---- the map stream (waves 4-7): slot h of the ring takes the 16 rows of half-step h, wave w rows 4 (w - 4) .. + 3, one
    //      row (P / 4 lanes x 16 B) per DMA instruction; half-steps past the end are out of range and arrive as zeros in a slot
    //      nobody reads (the request count per step stays constant: the counted waits depend on it)
    const bool dma_lane = lane < (P >> 2);
    auto dma_half = [&](int h) {
        if (dma_lane) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = (wave - 4) * 4 + j;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    f_rsrc, (__attribute__((address_space(3))) void*)(uintptr_t)(smem + (size_t)(h % NHS) * HSLOT + r * RB), 16,
                    lane * 16, (h * HROWS + r) * RB, 0, MG_IMG_AUX);
            }
        }
    };
    // ---- the W stream (waves 0-3): the 19 fragments (1 KiB each, fragment-major in memory) of k-step s -> W image s & 1; wave w
    //      takes fragments w, w + 4, ..: five requests each, the 20th is a dummy (out of range: zeros into the image's spare KB)
    auto dma_w = [&](int s) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int t = wave + 4 * j;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                w_rsrc, (__attribute__((address_space(3))) void*)(uintptr_t)(smem + OFF_W + (size_t)(s & 1) * WBUF + t * 1024), 16,
                lane * 16, t < NT && s < nks ? (t * nks + s) * 1024 : 0x7ffffc00, 0, 0);
        }
    };

    f32x4 acc[NR][NW];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int t = 0; t < NW; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // conversion task (threads of waves 0-3 only: the map waves carry eight DMA requests per step, the W waves five cheap ones, so
    // the conversion is the W waves' to keep the two sides of a SIMD level): (k-octet ko, region quad pq): 8 rows x 4 regions
    // -> four 16-B chunks of the operand image
    const int nquad = P >> 2;
    const bool cv_on = !STAGER && tid < 4 * nquad;
    const int ko = cv_on ? tid / nquad : 0, pq = cv_on ? tid - ko * nquad : 0;
    const unsigned cv_src = (unsigned)((ko & 1) * 8 * RB + pq * 16);         // inside half-slot ko >> 1 of the k-step
    // image: [p][4 chunks], chunk c of row p at slot (c + 2 (p >> 2)) & 3: conflict-free for the MFMA operand reads
    // (a ds_read_b128's 16-lane groups take rows 0-3 / 12-15 of chunk c and rows 4-11 of chunk c + 1)
    auto a_off = [](int p, int c) { return (unsigned)(p * 64 + (((c + 2 * (p >> 2)) & 3) << 4)); };
    const unsigned cv_dst = a_off(4 * pq, ko);           // rows 4 pq .. 4 pq + 3 share (p >> 2): consecutive 64-B rows, same slot
    // max-pool task: row tid >> 4 of the k-step, region quads (tid & 15) + 16 m
    const int pk = tid >> 4, psub = tid & 15;
    // MFMA operand reads: map fragment = row 16 (7 rg + i) + (lane & 15), chunk lane >> 4; W fragment t of this wave
    const unsigned a_rd = abuf + (unsigned)(rg * 7 * 16 * 64) + a_off(lane & 15, lane >> 4);
    const unsigned w_rd = wbuf + (unsigned)(cg * 5 * 1024 + lane * 16);

    float* const pp0 = pooled_part + ((size_t)b * 2) * K + pk;
    float* const pp1 = pp0 + K;
    float* const ppo = pooled ? pooled + (size_t)b * K + pk : nullptr;
    auto max3 = [](float a, float b2, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b2), "v"(c)); return r; };
    // ---- (a) fp32 half-slots of k-step s -> bf16 operand image A[s & 1] (W waves)
    auto convert = [&](int s) {
        if (!cv_on) return;
        const unsigned ab = abuf + (unsigned)((s & 1) * ABUF);
        const unsigned src = ring + (unsigned)(((2 * s + (ko >> 1)) % NHS) * HSLOT) + cv_src;
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (PT > 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[j]) : "v"(src), "n"(j * PT * 4) : "memory");
            else asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(src + j * RB) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            u32x4 c;
            c[0] = pack2(v[0][e], v[1][e]); c[1] = pack2(v[2][e], v[3][e]); c[2] = pack2(v[4][e], v[5][e]); c[3] = pack2(v[6][e], v[7][e]);
            asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(ab + cv_dst), "v"(c), "n"(e * 64) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // ---- (b) max-pool of the 32 rows of k-step s: row pk, quads psub + 16 m.  In pieces, so that it can be spread over the gaps
    //      of an MFMA stream: reads | first maxima | second maxima + the 16-lane reduction | stores
    f32x4 q[4];
    float pmx = 0.f;
    auto pool_reads = [&](int s) {
        const unsigned src = ring + (unsigned)(((2 * s + (pk >> 4)) % NHS) * HSLOT) + (unsigned)((pk & 15) * RB + psub * 16);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            // (a compile-time region count: the first quads of every thread exist, no branch around their reads, no default)
            if (!(PT > 0 && 16 * m + 15 < PT / 4)) q[m] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if ((PT > 0 && 16 * m + 15 < PT / 4) || psub + 16 * m < nquad)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[m]) : "v"(src), "n"(m * 256) : "memory");
        }
    };
    auto pool_max_a = [&]() { pmx = max3(max3(q[0][0], q[0][1], q[0][2]), max3(q[0][3], q[1][0], q[1][1]), max3(q[1][2], q[1][3], q[2][0])); };
    auto pool_max_b = [&]() {
        pmx = max3(pmx, max3(q[2][1], q[2][2], q[2][3]), max3(q[3][0], q[3][1], max3(q[3][2], q[3][3], q[3][3])));
        pmx = row16_max(pmx);
    };
    // after the 16-lane reduction every lane of a row holds the maximum: lanes 0 / 1 / 2 of the row store it to the two partial
    // arrays and (if asked for) the combined one -- ONE store instruction per wave and k-step (counted below)
    float* const pdst = psub == 0 ? pp0 : psub == 1 ? pp1 : ppo;
    const bool pst_on = psub < (pooled ? 3 : 2);
    auto pool_store = [&](int s) {
        if (pst_on) pdst[s * KROWS] = pmx;
    };
    // ---- (c) the MFMAs of k-step s (map fragments from A[s & 1], W fragments from W[s & 1]), and IN THEIR GAPS -- a SIMD
    //      issues about one other instruction in the shadow of a 16-cycle MFMA -- this wave's requests of the step and the
    //      max-pool of k-step s + 1 (PT > 0: the number of LDS reads in flight is known at compile time, which the counted
    //      waits for the operand fragments need; otherwise the pool runs behind the MFMAs)
    auto mfma_step = [&](int s, bool pool) {
        const unsigned ab = (unsigned)((s & 1) * ABUF), wb = (unsigned)((s & 1) * WBUF);
        u32x4 wf[NW], bf[NR];
#pragma unroll
        for (int t = 0; t < NW; ++t) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[t]) : "v"(w_rd + wb), "n"(t * 1024) : "memory");
#pragma unroll
        for (int i = 0; i < NR; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bf[i]) : "v"(a_rd + ab), "n"(i * 1024) : "memory");
        // LDS reads of a wave complete in order: row tile i's MFMAs wait for the W fragments and map fragments 0 .. i only
        static_for_img<0, NR>([&](auto ic) {
            constexpr int i = decltype(ic)::v;
            constexpr int NPR = 4;                       // pool reads behind the operand reads (PT > 0: all four are issued)
            if constexpr (PT > 0 && i == 1) {
                if (pool) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR - 2 + NPR) : "memory");
                else asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR - 2) : "memory");
            }
            else if constexpr (PT > 0 && i >= 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR - 1 - i) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 bv = __builtin_bit_cast(bf16x8, bf[i]);
#pragma unroll
            for (int t = 0; t < NW; ++t)
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[t]), bv, acc[i][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // the gap behind row tile i
            if constexpr (i == 0) {
                if (STAGER) dma_half(2 * s + NHS);       // the half-slots of k-step s were converted in the step before
                else dma_w(s + 1);                       // the other W image was read by the MFMAs of the step before
                if (PT > 0 && pool) pool_reads(s + 1);
            }
            if constexpr (i == 1) { if (STAGER) dma_half(2 * s + NHS + 1); }
            if constexpr (PT > 0 && i == 2) { if (pool) pool_max_a(); }
            if constexpr (PT > 0 && i == 3) { if (pool) pool_max_b(); }
            if constexpr (PT > 0 && i == 4) { if (pool) pool_store(s + 1); }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (PT == 0 && pool) {
            pool_reads(s + 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            pool_max_a();
            pool_max_b();
            pool_store(s + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int nst = 1;                                   // pool store instructions per wave and k-step
    // "My requests up to x have landed" is a COUNT (in-order completion): at most as many operations outstanding as this wave
    // has issued after them.  Request order of a step j -- map waves: [rows of half-steps 2j+7, 2j+8 (4 + 4)] [pool stores of
    // k-step j+1]; W waves: [5 fragments of k-step j+1] [pool stores of k-step j+1].
    if (STAGER) {
#pragma unroll
        for (int h = 0; h < NHS; ++h) dma_half(h);
        wait_vm(4 * (NHS - 2));                          // own rows of k-step 0 (half-steps 0, 1)
    } else {
        dma_w(0);
        wait_vm(0);
    }
    asm volatile("s_barrier" ::: "memory");              // ... everyone's
    convert(0);
    pool_reads(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    pool_max_a();
    pool_max_b();
    pool_store(0);
    __builtin_amdgcn_sched_barrier(0);
    if (STAGER) wait_vm(4 * (NHS - 4) + nst);            // own rows of k-step 1
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#ifdef MG_IMG_TRACE
    unsigned long long tt[6] = {0, 0, 0, 0, 0, 0};
#define MG_TT(i, t0) { const unsigned long long t1_ = IMG_T(); tt[i] += t1_ - t0; t0 = t1_; }
#else
#define MG_TT(i, t0)
#endif
    // Step s: the MFMAs of k-step s (with the step's requests and the pool of k-step s + 1 in their gaps) on every wave; the W
    // waves then convert k-step s + 1 into the other operand image; one barrier.
    for (int s = 0; s < nks; ++s) {
#ifdef MG_IMG_TRACE
        unsigned long long t0 = IMG_T();
#endif
        mfma_step(s, s + 1 < nks);
        MG_TT(2, t0);
        if (!STAGER && s + 1 < nks) convert(s + 1);
        MG_TT(1, t0);
        // map waves: own rows of k-step s + 2 (half-steps 2s+4, 2s+5; the latter requested first in step s - 1 or, s = 0, sixth
        // in the prologue): younger are 4 rows + the stores of that step and all of this step.  W waves: own fragments of
        // k-step s + 1: younger are this step's stores.
        wait_vm(STAGER ? 12 + 2 * nst : nst);
        MG_TT(3, t0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        MG_TT(4, t0);
    }
#ifdef MG_IMG_TRACE
    if (lane == 0 && (wave == 0 || wave == 4) && (blockIdx.x == 0 || blockIdx.x == 129)) {
        unsigned long long* g = g_img_trace[(blockIdx.x ? 2 : 0) + (wave >> 2)];
        for (int i = 0; i < 5; ++i) g[i] = tt[i];
    }
#endif
    // ---- epilogue: + bias, bf16, through LDS (over the ring: every request has landed), 16-B row stores; columns N..319 zero.
    // The tiles were computed TRANSPOSED (A operand = W fragment, B operand = map fragment): this lane's accumulator element
    // [i][t][r] is output column (5 cg + t) * 16 + 4 * (lane >> 4) + r of region row 16 (7 rg + i) + (lane & 15), i.e. four
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
            const int row = (rg * 7 + i) * 16 + (lane & 15);
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
        if ((wave & 3) == 3) img_wave<6, 4, false, PT>(smem, feat, b, K, P, Wp, bias, N, bank, pooled_part, pooled, wave, lane, tid);
        else img_wave<6, 5, false, PT>(smem, feat, b, K, P, Wp, bias, N, bank, pooled_part, pooled, wave, lane, tid);
    } else {
        if ((wave & 3) == 3) img_wave<7, 4, true, PT>(smem, feat, b, K, P, Wp, bias, N, bank, pooled_part, pooled, wave, lane, tid);
        else img_wave<7, 5, true, PT>(smem, feat, b, K, P, Wp, bias, N, bank, pooled_part, pooled, wave, lane, tid);
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
