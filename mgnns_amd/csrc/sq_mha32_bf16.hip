// Fused single-query multi-head attention, bf16 operands, on v_mfma_f32_32x32x16_bf16 (round 4).
// submodules.py:55-119 with len_q == 1: K / V projections of the memory bank (fp32 accumulation), scores, mask, softmax and
// the probability-weighted sum; K and V never leave the accumulators.
//
// Geometry.  Bank rows go in tiles of 32, the model dim in 19 k-steps of 16 (300 -> 304: the 16x16x32 form padded to 320).
// A workgroup (8 waves) stages its rows ONCE in LDS by LDS-DMA, 624-B row stride (39 chunks of 16 B, odd: the B/A-fragment
// pattern row = lane & 31, chunk = lane >> 5 of ds_read_b128 is conflict free).  Work is cut into UNITS = (head, K or V) x
// 32 head dims; the two waves of a 32-dim slice (they share a SIMD, hence its matrix pipe) draw the slice's units from an LDS
// ticket counter (K(h0) K(h1) V(h0) V(h1) K(h2) ...), hand-over through LDS counters, no s_barrier after staging.
//   K units are computed TRANSPOSED  (D = W X^T: lane = bank row, registers = 16 head dims): the score reduction over the
//           head dims is in-register + ONE v_permlane32_swap per pair of row tiles;
//   V units are computed STRAIGHT    (D = X W^T: lane = head dim, registers = 16 bank rows): the probability-weighted sum over
//           the rows is in-register (probabilities are LDS broadcasts) + one half swap; 128-B coalesced store of `o`.
// Both use the SAME bank and weight fragments (the A and B operand layouts of the instruction coincide): only the operand
// order of the MFMA differs.
//
// Two kernels:
//   sq_mha32_core_kernel    one workgroup per sample (x head pairs for small batches): the image banks (L = 196, no mask) --
//                           the launch the MFMA-utilisation figure is quoted on -- and any masked bank without a plan;
//   sq_mha32_packed_kernel  masked text banks: the batch's live rows are PACKED -- samples of a few tokens share a workgroup
//                           (8-row aligned, <= 128 rows, <= 16 samples per group; plan by sq_mha32_plan_kernel once per batch),
//                           so the weight stream is read once per 128 live rows instead of once per sample and one 100-token
//                           document no longer holds a 256-CU launch open (MODEL:509-527: mean 16 tokens of T = 100).
//                           Scores use the row's own sample's query, the softmax is segmented per sample, the weighted sum
//                           is taken per 8-row block in registers and the blocks of a sample are added through LDS.
#include "common.hpp"
#include "sq_mha_plan.hpp"

#ifdef MG_MHA32_TRACE
// profiling aid (off by default; tools/dev/mha32_trace.py): s_memtime stamps of waves 0 (K units) and 4 (V units) of workgroups
// 0 and 129 of sq_mha32_core_kernel at every phase boundary
__device__ unsigned long long g_mha32_trace[4][64];
#define MG_STAMP(slot)                                                                              \
    do {                                                                                            \
        if ((threadIdx.x & 255) == 0 && (blockIdx.x == 0 || blockIdx.x == 129) && blockIdx.y == 0)  \
            g_mha32_trace[(blockIdx.x ? 2 : 0) + (threadIdx.x >> 8)][(slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define MG_STAMP(slot) do { } while (0)
#endif

namespace {

typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));
typedef int i32x4v __attribute__((ext_vector_type(4)));

constexpr int RT = 32;                  // bank rows per tile
constexpr int MT = 7;                   // row tiles a workgroup can hold (L <= 224)
constexpr int LMAX = MT * RT;
constexpr int KS = 19;                  // k-steps of 16 over the model dim (300 -> 304)
constexpr int KCH = 2 * KS;             // 16-byte chunks of a bank row that are used
constexpr int CH = 40;                  // chunks per bank row in HBM ([.., 320] bf16)
constexpr int LSTR = 39;                // LDS row stride in chunks (624 B)
constexpr int DK = 128;
constexpr int NTHR = 512;
constexpr int FRAG = 1024;              // bytes per weight fragment (64 lanes x 16 B)
constexpr int QMAX = 2048;              // floats of projected query a per-sample workgroup keeps (H * 128 <= QMAX)
constexpr int MAXH = QMAX / DK;
constexpr int TILE_BYTES = RT * LSTR * 16;

__device__ __forceinline__ unsigned short f2bf(float x) {      // round-to-nearest-even
    unsigned int u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// Wp[h][kv][slice][ks][lane][8] = W_kv[h*128 + slice*32 + (lane&31)][ks*16 + (lane>>5)*8 + j]  (0 beyond D)
__global__ __launch_bounds__(256) void pack_kv_weights32_kernel(const float* __restrict__ Wk, const float* __restrict__ Wv,
                                                                int H, int D, unsigned short* __restrict__ Wp) {
    const size_t total = (size_t)H * 2 * 4 * KS * 64;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        size_t r = i >> 6;
        const int ks = (int)(r % KS); r /= KS;
        const int sl = (int)(r & 3); r >>= 2;
        const int kv = (int)(r & 1);
        const int h = (int)(r >> 1);
        const float* W = kv ? Wv : Wk;
        const int row = h * DK + sl * 32 + (lane & 31);
        const int k0 = ks * 16 + (lane >> 5) * 8;
        unsigned short v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (k0 + j < D) ? f2bf(W[(size_t)row * D + k0 + j]) : (unsigned short)0;
        uint4 o;
        o.x = v[0] | ((unsigned)v[1] << 16);
        o.y = v[2] | ((unsigned)v[3] << 16);
        o.z = v[4] | ((unsigned)v[5] << 16);
        o.w = v[6] | ((unsigned)v[7] << 16);
        reinterpret_cast<uint4*>(Wp)[i] = o;
    }
}

// v_permlane32_swap exchanges the upper half of its first operand with the lower half of its second: afterwards
// a = [a.lo, b.lo], b = [a.hi, b.hi], so a + b = [a.lo + a.hi | b.lo + b.hi]: ONE swap + one add folds the two 32-lane halves of
// TWO values.  Inline asm: both registers are read AND written; the s_nop cover the VALU-write -> swap-read and swap-write ->
// VALU-read hazards the compiler's hazard recogniser does not see through an asm block.
__device__ __forceinline__ float halves_sum2(float a, float b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

// LDS counters instead of s_barrier (LDS operations of a wave complete in order: lgkmcnt(0) in front of an arrival publishes
// this wave's LDS writes to whoever sees the count)
__device__ __forceinline__ int lds_arrive(int* ctr, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __builtin_amdgcn_readfirstlane(old);
}
__device__ __forceinline__ void lds_wait_ge(int* ctr, int target) {
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
        __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

template <int N> struct IC { static constexpr int v = N; };
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(IC<I>{});
        static_for<I + 1, N>(f);
    }
}

// Weight fragments through a buffer resource: one VGPR (lane * 16) addresses every fragment, the fragment is a wave-uniform
// byte offset in an SGPR
struct WStream {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;
};
__device__ __forceinline__ uint4 wfrag(const WStream& w, int soff) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(w.rsrc, w.voff, soff, 0));
}

#ifndef MG_MHA32_RING
#define MG_MHA32_RING 8
#endif
#ifndef MG_MHA32_BD7
#define MG_MHA32_BD7 3       // (4: the two-branch unit loop spills a next-unit weight fragment behind a vmcnt(0))
#endif
#ifndef MG_MHA32_GROUP
#define MG_MHA32_GROUP 2
#endif


// Fragments in flight.  bq: the weight fragments of the next BD k-steps (L2 resident, ~1 us away under load: the fewer row tiles
// a k-step has, the more k-steps run ahead); they stay in flight ACROSS a unit boundary -- k-step j of a unit is expected in
// slot j % BD, and the last BD k-steps of a unit refill their slot with the NEXT unit's k-step ks % BD (the slot the next unit
// looks for it in; BD need not divide 19).
template <int NT>
struct Frags {
    static constexpr int BD = NT >= 7 ? MG_MHA32_BD7 : NT >= 4 ? 6 : NT >= 2 ? 10 : KS;
    static constexpr int TOTAL = KS * NT;
    static constexpr int RA = TOTAL < MG_MHA32_RING ? TOTAL : MG_MHA32_RING;
    uint4 bq[BD];
};
template <int NT>
__device__ __forceinline__ void frags_prime(Frags<NT>& f, const WStream& w, int wb) {
#pragma unroll
    for (int d = 0; d < Frags<NT>::BD; ++d) f.bq[d] = wfrag(w, wb + d * FRAG);
}

// acc[i] += (VT ? X_i W^T : W X_i^T) over the padded model dim for NT row tiles of 32.  Hand-scheduled like the 16x16x32 form
// it replaces: bank fragments by inline-asm ds_read_b128 from two base registers with immediate offsets (tiles 0-3 / 4-6) into
// a ring that runs RA tiles ahead, ONE counted s_waitcnt per GRP row tiles, the refill of a ring slot right behind the MFMA
// that consumed it; sched_barrier(0) fences pin the order.  The whole GEMM is ONE basic block (wb_next, the next unit's weight
// stream, is computed by the caller: a branch in the middle made hipcc spill the ring fragments that were live across it).
// One LDS ticket (ds_add_rtn_u32 by lane 0 only) WITHOUT a branch: the exec mask is narrowed inside the asm statement, so the
// draw can sit in the middle of the GEMM's basic block.  The result lands like any LDS read (in order): read it with
// lds_ticket_value() behind a wait that covers it.
__device__ __forceinline__ int lds_ticket_issue(unsigned addr) {
    int ret;
    unsigned long long sv;
    const int one = 1;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tds_add_rtn_u32 %0, %2, %3\n\ts_mov_b64 exec, %1"
                 : "=&v"(ret), "=&s"(sv) : "v"(addr), "v"(one) : "memory");
    return ret;
}
template <int N>
__device__ __forceinline__ int lds_ticket_value(int raw) {          // s_waitcnt lgkmcnt(N), then lane 0's value as a scalar
    int t;
    asm volatile("s_waitcnt lgkmcnt(%2)\n\tv_readfirstlane_b32 %0, %1" : "=s"(t) : "v"(raw), "n"(N) : "memory");
    return t;
}

// LATE (the 7-tile class): the next unit's ticket is drawn INSIDE the GEMM, two k-steps before its weight stream is first needed
// (k-step KS - BD - 2: issue; k-step KS - BD: value -> next_stream(t) gives the stream's byte offset), so the faster wave of a
// slice takes the next unit when it gets there -- drawn a whole unit ahead, the slower wave's reserved unit held everybody up
// (tools/dev/mha32_trace.py: epilogues of 5-8 k cycles waiting for it).  !LATE: wb_next is the caller's.
template <int NT, bool VT, bool LATE, typename NextStream>
__device__ __forceinline__ void kv_gemm(f32x16 (&acc)[NT], Frags<NT>& f, unsigned a_lo, unsigned a_hi, const WStream& w, int wb,
                                        int wb_next, unsigned ticket_addr, NextStream&& next_stream) {
    constexpr int BD = Frags<NT>::BD, RA = Frags<NT>::RA, TOTAL = Frags<NT>::TOTAL;
    constexpr int GRP = MG_MHA32_GROUP < RA ? MG_MHA32_GROUP : 1;
    static_assert(!LATE || KS - BD - 2 >= 0, "late draw needs two k-steps in front of the first next-unit fragment");
    u32x4 ga[RA];
    int raw_ticket = 0;
    auto fetch = [&](auto nc) {
        constexpr int n = decltype(nc)::v;
        constexpr int ks = n / NT, i = n % NT;
        if constexpr (i < 4) ga[n % RA] = mg_lds_read128<i * TILE_BYTES + ks * 32>(a_lo);
        else ga[n % RA] = mg_lds_read128<(i - 4) * TILE_BYTES + ks * 32>(a_hi);
    };
    static_for<0, RA>(fetch);
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, KS>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::v;
        const bfx8 wv = __builtin_bit_cast(bfx8, f.bq[ks % BD]);
        if constexpr (LATE && ks == KS - BD) {
            // >= 2 NT fragment reads were issued behind the draw and at most RA are in flight: it has landed
            wb_next = next_stream(lds_ticket_value<RA>(raw_ticket));
            __builtin_amdgcn_sched_barrier(0);
        }
        static_for<0, (NT + GRP - 1) / GRP>([&](auto gc) {
            constexpr int i0 = decltype(gc)::v * GRP;
            constexpr int cnt = i0 + GRP <= NT ? GRP : NT - i0;
            constexpr int n0 = ks * NT + i0;
            constexpr int issued = n0 + RA < TOTAL ? n0 + RA : TOTAL;
            // (LATE: between the draw and its read one more LDS operation is in flight, OLDER than every fragment requested after
            //  it: the count below then waits for one fragment more than needed, never for fewer)
            mg_lds_wait<issued - n0 - cnt>();
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, cnt>([&](auto jc) {
                constexpr int i = i0 + decltype(jc)::v;
                const bfx8 xv = __builtin_bit_cast(bfx8, ga[(ks * NT + i) % RA]);
                if constexpr (VT) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xv, wv, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, xv, acc[i], 0, 0, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, cnt>([&](auto jc) {
                constexpr int n2 = n0 + decltype(jc)::v + RA;
                if constexpr (n2 < TOTAL) fetch(IC<n2>{});
            });
            if constexpr (LATE && ks == KS - BD - 2 && i0 == 0) raw_ticket = lds_ticket_issue(ticket_addr);
            __builtin_amdgcn_sched_barrier(0);
        });
        if (ks + BD < KS) f.bq[ks % BD] = wfrag(w, wb + (ks + BD) * FRAG);
        else f.bq[ks % BD] = wfrag(w, wb_next + (ks % BD) * FRAG);        // the next unit's k-step ks % BD
        __builtin_amdgcn_sched_barrier(0);
    });
    // the accumulators are "used" HERE: MFMAs are pure to the compiler, and with the first use of their results behind a wait
    // loop (the epilogues' hand-over waits) it sinks all of them past the loop -- away from the fragment reads they were
    // interleaved with, every ring fragment spilled on the way
#pragma unroll
    for (int i = 0; i < NT; ++i) asm volatile("" : "+v"(acc[i]));
}

template <int NT>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[NT]) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
}

// ---- LDS-DMA staging: rows_live rows of 38 chunks into the padded [row][39 chunk] image, walked linearly (a DMA instruction
//      fills 1 KiB of CONTIGUOUS LDS from per-lane global addresses); lanes on the pad chunk of a row are switched off.
//      src(row) -> the row's first chunk in HBM, or nullptr for a row that reads as zeros (then the 16 zero bytes `zero`).
template <typename RowSrc>
__device__ __forceinline__ void stage_pieces(unsigned char* smem, int rows_live, const uint4* zero, RowSrc&& src, int pc0, int pc1,
                                             int first, int step) {
    // pieces [pc0, pc1) of the image (1 KiB each), this wave taking pc0 + first, + step, ...
    const int lane = threadIdx.x & 63;
    const int total = rows_live * LSTR;
    for (int pc = pc0 + first; pc < pc1 && pc * 64 < total; pc += step) {
        const int g = pc * 64 + lane;
        const int row = g / LSTR, c = g - row * LSTR;
        if (g < total && c < KCH) {
            const uint4* r0 = src(row);
            const uint4* s = r0 ? r0 + c : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                             (__attribute__((address_space(3))) void*)(uintptr_t)(smem + (size_t)pc * 1024), 16, 0, 0);
        }
    }
}
template <typename RowSrc>
__device__ __forceinline__ void stage_rows(unsigned char* smem, int rows_live, const uint4* zero, RowSrc&& src) {
    stage_pieces(smem, rows_live, zero, src, 0, 0x7fffffff, (int)(threadIdx.x >> 6), NTHR / 64);
}

// =====================================================================================================================
// one workgroup per sample
// =====================================================================================================================
constexpr int NSLOT_P = 8;                                               // probability rows kept (heads between softmax and weighted sum)
constexpr size_t OFF_PART = (size_t)LMAX * LSTR * 16;                    // float [2][4][LMAX] partial scores (head parity, slice)
constexpr size_t OFF_P = OFF_PART + 2 * 4 * LMAX * sizeof(float);       // float [NSLOT_P][LMAX] probabilities
constexpr size_t OFF_MB = OFF_P + NSLOT_P * LMAX * sizeof(float);       // float [LMAX] mask bias: 0 or -inf
constexpr size_t OFF_INT = OFF_MB + LMAX * sizeof(float);               // int [16 + 3 * MAXH]: live rows, tickets, arrival counts
constexpr size_t OFF_Q = OFF_INT + (16 + 3 * MAXH) * sizeof(int);       // float [QMAX] this sample's projected query
constexpr size_t SMEM_BYTES = OFF_Q + QMAX * sizeof(float);
static_assert(SMEM_BYTES <= 160 * 1024, "LDS");
static_assert(OFF_Q % 16 == 0 && OFF_P % 16 == 0 && OFF_PART % 16 == 0, "LDS alignment");

// Everything after the bank is staged, for a compile-time tile-count class.
//
// Work is a QUEUE of units per slice: the two waves of a slice (w and w + 4: they share a SIMD, hence its matrix pipe) draw
// the slice's units from an LDS ticket counter in the order K(h0) K(h1) V(h0) V(h1) K(h2) ..., each wave holding its current
// unit and the next one (drawn a unit ahead, so that the next unit's weight stream is known before the GEMM starts and the GEMM
// stays one basic block).  The older wave of a SIMD gets ~90 % of a contended matrix pipe: with FIXED roles (K units on wave w,
// V units on wave w + 4: measured, tools/dev/mha32_trace.py) the K wave was done after 79 k cycles and the V wave ran its last
// three units alone until 106 k; with the queue the faster wave simply takes more units and both finish within a unit of each
// other, the pipe busy with one wave's MFMAs while the other is in its VALU-only epilogue.
// No s_barrier after the staging one; hand-over through LDS counters:
//   K unit   GEMM -> this slice's partial scores -> arrival at the head's count; the LAST of the four slices to arrive runs the
//            head's softmax (one wave, four positions per lane) and publishes the probabilities
//   V unit   GEMM -> wait for the head's probabilities (published a whole GEMM earlier, as a rule) -> weighted sum -> o
// Partial scores of local head n reuse the slot of head n - 2 (wait for its softmax), probabilities of head n the row of head
// n - 8 (wait for its four weighted sums).  A unit only ever waits for LOWER tickets, and the lowest outstanding ticket of a
// slice is always some wave's current unit: no deadlock.
template <int NT>
__device__ __forceinline__ void mha_body(unsigned char* smem, int B, int L, int H, const unsigned short* __restrict__ Wp,
                                         const float* __restrict__ bv, float temp, float* __restrict__ o,
                                         float* __restrict__ attn, int lvalid) {
    uint4* Xs = reinterpret_cast<uint4*>(smem);
    float* s_part = reinterpret_cast<float*>(smem + OFF_PART);
    float* s_p = reinterpret_cast<float*>(smem + OFF_P);
    const float* s_mb = reinterpret_cast<const float*>(smem + OFF_MB);
    int* s_int = reinterpret_cast<int*>(smem + OFF_INT);
    int* s_ticket = s_int + 4;
    int* s_kdone = s_int + 16;
    int* s_smdone = s_int + 16 + MAXH;
    int* s_pvdone = s_int + 16 + 2 * MAXH;
    const float* s_q = reinterpret_cast<const float*>(smem + OFF_Q);
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave & 3;                                        // slice: head dims 32 wq ... 32 wq + 31
    const int b = blockIdx.x;
    const unsigned a_lo = mg_lds_addr(Xs + (lane & 31) * LSTR + half), a_hi = a_lo + 4 * TILE_BYTES;
    const float inv_temp = 1.0f / temp;
    WStream wsr;
    wsr.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Wp), 0, 0x7fffffff, 0x00027000);
    wsr.voff = lane * 16;
    // this workgroup's head pairs: blockIdx.y, + gridDim.y, ...: local head n = 2 (pair index) + (0 | 1); tickets per slice in the
    // order K(2p) K(2p+1) V(2p) V(2p+1) per pair p -- and K(2p) V(2p) for a last pair with ONE head (odd H): every ticket below
    // nunits is a unit (no skipped tickets: a draw is one atomic, branch free)
    const int gy = (int)gridDim.y, y = (int)blockIdx.y;
    const int pairs = (H + 1) / 2;
    const int npairs = (pairs - y + gy - 1) / gy;
    const int last_h0 = (y + (npairs - 1) * gy) * 2;                // first head of this workgroup's last pair
    const int nheads = npairs * 2 - (last_h0 + 1 >= H ? 1 : 0);
    const int nunits = 2 * nheads;
    auto unit_of = [&](int t, int& n, bool& vunit) {                // ticket -> (local head, K or V); anything for t >= nunits
        const int p = t >> 2, j = t & 3;
        const bool single = 2 * p + 1 >= nheads;
        n = 2 * p + (single ? 0 : (j & 1));
        vunit = single ? (j & 1) != 0 : (j >> 1) != 0;
    };
    auto head_of = [&](int n) { return (y + (n >> 1) * gy) * 2 + (n & 1); };
    const unsigned ticket_addr = mg_lds_addr(s_ticket + wq);
    auto draw = [&]() {
        int v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(s_ticket + wq, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __builtin_amdgcn_readfirstlane(v);
    };
    auto wstream = [&](int t) {                         // byte offset of the unit's first fragment; past the end: a harmless re-read
        int n;
        bool v;
        unit_of(t < nunits ? t : 0, n, v);
        return (((head_of(n) * 2 + (v ? 1 : 0)) * 4 + wq) * KS) * FRAG;
    };
    const __amdgpu_buffer_rsrc_t attn_rsrc = __builtin_amdgcn_make_buffer_rsrc(attn, 0, attn ? 0x7fffffff : 0, 0x00027000);

    int t = draw();
    Frags<NT> f;
    frags_prime<NT>(f, wsr, wstream(t));                // weight fragments on their way while the bank DMA lands
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's LDS-DMA pieces of the bank (not tracked by hipcc)
    MG_STAMP(1);
    __syncthreads();                                    // ... and every other wave's
    MG_STAMP(2);
    int stamp = 3;
    (void)stamp;

    constexpr bool LATE = NT >= MT;                     // (the small classes run their weight fragments a whole unit ahead)
    while (t < nunits) {
        int t_next = LATE ? nunits : draw();            // !LATE: drawn a unit ahead (its latency hides behind the GEMM)
        int n;
        bool vunit;
        unit_of(t, n, vunit);
        const int h = head_of(n);
        const int wb = wstream(t), wb_next = LATE ? 0 : wstream(t_next);
        auto next_stream = [&](int tn) { t_next = tn; return wstream(tn); };
        f32x16 acc[NT];
        acc_zero<NT>(acc);
        if (!vunit) {
            kv_gemm<NT, false, LATE>(acc, f, a_lo, a_hi, wsr, wb, wb_next, ticket_addr, next_stream);
            MG_STAMP(stamp++);
            // ---- partial scores of this slice's 32 head dims: lane = bank row, registers = dims (r&3) + 8 (r>>2) + 4 half.
            //      q.(K_l + b_k) = q.K_l + const: b_k never enters (softmax invariant).
            const float* qv = s_q + h * DK + wq * 32 + 4 * half;
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(qv), q1 = *reinterpret_cast<const f32x4*>(qv + 8),
                        q2 = *reinterpret_cast<const f32x4*>(qv + 16), q3 = *reinterpret_cast<const f32x4*>(qv + 24);
            float v[(NT + 1) / 2 * 2];
#pragma unroll
            for (int i = 0; i < (NT + 1) / 2 * 2; ++i) {
                float a = 0.f, c = 0.f;
                if (i < NT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        a = fmaf(q0[r], acc[i][r], a);
                        c = fmaf(q1[r], acc[i][4 + r], c);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        a = fmaf(q2[r], acc[i][8 + r], a);
                        c = fmaf(q3[r], acc[i][12 + r], c);
                    }
                }
                v[i] = a + c;
            }
            if (n >= 2) lds_wait_ge(s_smdone + n - 2, 1);           // the slot's previous head has been consumed
            float* part = s_part + ((n & 1) * 4 + wq) * LMAX;
#pragma unroll
            for (int g = 0; g < (NT + 1) / 2; ++g) {
                const float s2 = halves_sum2(v[2 * g], v[2 * g + 1]);      // lanes 0-31: tile 2g, lanes 32-63: tile 2g + 1
                if (2 * g + 1 < NT || lane < 32) part[64 * g + lane] = s2;
            }
            if (lds_arrive(s_kdone + n, lane) == 3) {
                // ---- last slice of the head: masked softmax over all positions, four per lane
                if (n >= NSLOT_P) lds_wait_ge(s_pvdone + n - NSLOT_P, 4);
                const float* sp = s_part + (n & 1) * 4 * LMAX;
                float sc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pos = lane + 64 * j;
                    sc[j] = -INFINITY;
                    if (pos < lvalid)
                        sc[j] = ((sp[pos] + sp[pos + LMAX]) + (sp[pos + 2 * LMAX] + sp[pos + 3 * LMAX])) * inv_temp + s_mb[pos];
                }
                const float m = wave_max_dpp(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])));
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = (sc[j] != -INFINITY) ? __expf(sc[j] - m) : 0.f;
                const float z = wave_sum_dpp((e[0] + e[1]) + (e[2] + e[3]));
                const float rz = 1.0f / z;              // all masked: 0 * inf = NaN, like the reference's softmax of -inf
                float* prow = s_p + (n & (NSLOT_P - 1)) * LMAX;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pos = lane + 64 * j;
                    const float p = e[j] * rz;
                    if (pos < LMAX) prow[pos] = p;
                    if (attn && pos < L)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, p), attn_rsrc, pos * 4, (h * B + b) * L * 4, 0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(s_smdone + n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            MG_STAMP(stamp++);
        } else {
            kv_gemm<NT, true, LATE>(acc, f, a_lo, a_hi, wsr, wb, wb_next, ticket_addr, next_stream);
            MG_STAMP(stamp++);
            // ---- o[d] = sum_l p[l] (V[l,d] + bv[d]) = sum_l p[l] V[l,d] + bv[d]: lane = head dim, registers = bank rows
            //      (r&3) + 8 (r>>2) + 4 half of the tile; the probabilities of a tile are four 16-byte LDS broadcasts
            lds_wait_ge(s_smdone + n, 1);
            const float* pp = s_p + (n & (NSLOT_P - 1)) * LMAX + 4 * half;
            float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 pv = *reinterpret_cast<const f32x4*>(pp + 32 * i + 8 * g);
                    t0 = fmaf(pv[0], acc[i][4 * g + 0], t0);
                    t1 = fmaf(pv[1], acc[i][4 * g + 1], t1);
                    t2 = fmaf(pv[2], acc[i][4 * g + 2], t2);
                    t3 = fmaf(pv[3], acc[i][4 * g + 3], t3);
                }
            }
            if (nheads > NSLOT_P) lds_arrive(s_pvdone + n, lane);           // (only then is the row ever reused)
            const float th = (t0 + t1) + (t2 + t3);
            const float tot = halves_sum2(th, th);                          // both halves: the sum over all rows
            if (lane < 32) {
                const int d = h * DK + wq * 32 + lane;
                o[(size_t)b * H * DK + d] = tot + (bv ? bv[d] : 0.f);
            }
            MG_STAMP(stamp++);
        }
        t = t_next;
    }
}

__global__ __launch_bounds__(NTHR) void sq_mha32_core_kernel(const float* __restrict__ qh, const unsigned short* __restrict__ bank,
                                                             const float* __restrict__ mask, int B, int L, int H,
                                                             const unsigned short* __restrict__ Wp, const float* __restrict__ bv,
                                                             float temp, float* __restrict__ o, float* __restrict__ attn) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* s_lvalid = reinterpret_cast<int*>(smem + OFF_INT);
    float* s_mb = reinterpret_cast<float*>(smem + OFF_MB);
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const uint4* xb = reinterpret_cast<const uint4*>(bank) + (size_t)b * L * CH;
    MG_STAMP(0);

    // ---- live rows, mask bias, the unit queue's counters
    if (tid < 16 + 3 * MAXH) s_lvalid[tid] = (tid == 0 && !mask) ? L : 0;
    __syncthreads();
    {
        int last = 0;
        for (int t = tid; t < LMAX; t += NTHR) {
            const bool live = t < L && (!mask || mask[(size_t)b * L + t] != 0.0f);
            if (mask && live) last = t + 1;
            s_mb[t] = (t < L && !live) ? -INFINITY : 0.0f;
        }
        if (last) atomicMax(s_lvalid, last);
        {                               // this sample's query row -> LDS (visible behind the barrier)
            float* s_q = reinterpret_cast<float*>(smem + OFF_Q);
            for (int i = tid * 4; i < H * DK; i += NTHR * 4)
                *reinterpret_cast<f32x4*>(s_q + i) = *reinterpret_cast<const f32x4*>(qh + (size_t)b * H * DK + i);
        }
        __syncthreads();
    }
    const int lvalid = *s_lvalid;
    const int n_mt = (lvalid + RT - 1) / RT;
    const int n_sel = n_mt <= 1 ? 1 : n_mt <= 2 ? 2 : n_mt <= 4 ? 4 : MT;
    // rows >= L of the class read the zero padding at the end of bank row 0 (columns 312..319)
    stage_rows(smem, n_sel * RT, xb + (CH - 1), [&](int row) { return row < L ? xb + (size_t)row * CH : (const uint4*)nullptr; });
#ifdef MG_MHA32_ONLY                   // measurement builds: one tile-count class (register / code-size studies)
    mha_body<MG_MHA32_ONLY>(smem, B, L, H, Wp, bv, temp, o, attn, lvalid);
    return;
#endif
    switch (n_sel) {
        case 1: mha_body<1>(smem, B, L, H, Wp, bv, temp, o, attn, lvalid); break;
        case 2: mha_body<2>(smem, B, L, H, Wp, bv, temp, o, attn, lvalid); break;
        case 4: mha_body<4>(smem, B, L, H, Wp, bv, temp, o, attn, lvalid); break;
        default: mha_body<MT>(smem, B, L, H, Wp, bv, temp, o, attn, lvalid); break;
    }
}

// =====================================================================================================================
// packed masked banks
// =====================================================================================================================
using mg_plan::PR;                      // rows a group holds (4 tiles)
using mg_plan::PS;                      // samples a group holds
constexpr int PHL = 2;                  // heads a workgroup owns (one pair)
constexpr int PQS = PHL * DK + 4;       // floats per sample in the query image (+4: samples on different LDS banks)
using mg_plan::PLAN_HDR;

// (plan layout and construction: sq_mha_plan.hpp)
__global__ __launch_bounds__(1024) void sq_mha32_plan_kernel(const float* __restrict__ mask, int B, int L, int* __restrict__ plan) {
    extern __shared__ int s_plan[];
    mg_plan::build<1024>(mask, B, L, plan, s_plan);
}

constexpr size_t P_OFF_PART = (size_t)PR * LSTR * 16;                         // float [PHL][4][PR] partial scores
constexpr size_t P_OFF_P = P_OFF_PART + PHL * 4 * PR * sizeof(float);        // float [PHL][PR] scores -> probabilities
constexpr size_t P_OFF_MB = P_OFF_P + PHL * PR * sizeof(float);              // float [PR] mask bias
constexpr size_t P_OFF_SRC = P_OFF_MB + PR * sizeof(float);                  // int [PR] bank row (b * L + pos) or -1
constexpr size_t P_OFF_RSL = P_OFF_SRC + PR * sizeof(int);                   // int [PR] group-local sample of the row
constexpr size_t P_OFF_INT = P_OFF_RSL + PR * sizeof(int);                   // int [64]: tickets [4], kdone [2], smdone [2], soff [16], slen [16], group header
constexpr size_t P_OFF_Q = P_OFF_INT + 64 * sizeof(int);                     // float [PS][PQS] projected queries of the group's samples
constexpr size_t P_OFF_BLK = P_OFF_Q + PS * PQS * sizeof(float);             // float [PHL][4][PR / 8][32] weighted sums per 8-row block
constexpr size_t P_SMEM_BYTES = P_OFF_BLK + PHL * 4 * (PR / 8) * 32 * sizeof(float);
static_assert(P_SMEM_BYTES <= 160 * 1024, "LDS");
static_assert(P_OFF_Q % 16 == 0 && P_OFF_P % 16 == 0 && P_OFF_BLK % 16 == 0, "LDS alignment");

// A group's units for this workgroup's heads: the unit queue of mha_body (tickets K(h0) [K(h1)] V(h0) [V(h1)] per slice)
template <int NT>
__device__ __forceinline__ void packed_body(unsigned char* smem, int B, int L, int H, int h0, int hl, int first, int cnt,
                                            const unsigned short* __restrict__ Wp, const float* __restrict__ bv, float temp,
                                            float* __restrict__ o, float* __restrict__ attn) {
    uint4* Xs = reinterpret_cast<uint4*>(smem);
    float* s_part = reinterpret_cast<float*>(smem + P_OFF_PART);
    float* s_p = reinterpret_cast<float*>(smem + P_OFF_P);
    const float* s_mb = reinterpret_cast<const float*>(smem + P_OFF_MB);
    const int* s_rsl = reinterpret_cast<const int*>(smem + P_OFF_RSL);
    int* s_int = reinterpret_cast<int*>(smem + P_OFF_INT);
    int* s_ticket = s_int;
    int* s_kdone = s_int + 4;
    int* s_smdone = s_int + 6;
    const int* s_soff = s_int + 8;
    const int* s_slen = s_int + 24;
    const float* s_q = reinterpret_cast<const float*>(smem + P_OFF_Q);
    float* s_blk = reinterpret_cast<float*>(smem + P_OFF_BLK);
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave & 3;
    const unsigned a_lo = mg_lds_addr(Xs + (lane & 31) * LSTR + half), a_hi = a_lo + 4 * TILE_BYTES;
    WStream wsr;
    wsr.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Wp), 0, 0x7fffffff, 0x00027000);
    wsr.voff = lane * 16;
    const int nunits = 2 * hl;
    const float inv_temp = 1.0f / temp;
    auto draw = [&]() {
        int v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(s_ticket + wq, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __builtin_amdgcn_readfirstlane(v);
    };
    auto wstream = [&](int t) {                         // ticket t < hl: K of local head t; else V of local head t - hl
        const int tt = t < nunits ? t : 0;
        const int kv = tt >= hl ? 1 : 0;
        return ((((h0 + tt - kv * hl) * 2 + kv) * 4 + wq) * KS) * FRAG;
    };

    int t = draw();
    Frags<NT> f;
    frags_prime<NT>(f, wsr, wstream(t));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's LDS-DMA pieces
    __syncthreads();

    while (t < nunits) {
        const int t_next = draw();
        const int wb = wstream(t), wb_next = wstream(t_next);
        f32x16 acc[NT];
        acc_zero<NT>(acc);
        if (t < hl) {
            const int n = t, h = h0 + n;
            kv_gemm<NT, false, false>(acc, f, a_lo, a_hi, wsr, wb, wb_next, 0u, [](int) { return 0; });
            // ---- partial scores: the row's OWN sample's query (lane = packed row)
            float v[(NT + 1) / 2 * 2];
#pragma unroll
            for (int i = 0; i < (NT + 1) / 2 * 2; ++i) {
                float a = 0.f, c = 0.f;
                if (i < NT) {
                    const float* qv = s_q + s_rsl[32 * i + (lane & 31)] * PQS + n * DK + wq * 32 + 4 * half;
                    const f32x4 q0 = *reinterpret_cast<const f32x4*>(qv), q1 = *reinterpret_cast<const f32x4*>(qv + 8),
                                q2 = *reinterpret_cast<const f32x4*>(qv + 16), q3 = *reinterpret_cast<const f32x4*>(qv + 24);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        a = fmaf(q0[r], acc[i][r], a);
                        c = fmaf(q1[r], acc[i][4 + r], c);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        a = fmaf(q2[r], acc[i][8 + r], a);
                        c = fmaf(q3[r], acc[i][12 + r], c);
                    }
                }
                v[i] = a + c;
            }
            float* part = s_part + (n * 4 + wq) * PR;
#pragma unroll
            for (int g = 0; g < (NT + 1) / 2; ++g) {
                const float s2 = halves_sum2(v[2 * g], v[2 * g + 1]);
                if (2 * g + 1 < NT || lane < 32) part[64 * g + lane] = s2;
            }
            if (lds_arrive(s_kdone + n, lane) == 3) {
                // ---- last slice of the head: the softmax of every sample of the group over ITS rows; four samples at a time,
                //      one per 16-lane row (DPP reductions stay inside a row)
                const float* sp = s_part + n * 4 * PR;
                float* prow = s_p + n * PR;
                const int sub = lane & 15;
                for (int s0 = 0; s0 < cnt; s0 += 4) {
                    const int sl = s0 + (lane >> 4);
                    const bool act = sl < cnt;
                    const int off = act ? s_soff[sl] : 0, lv = act ? s_slen[sl] : 0;
                    const int lv0 = __builtin_amdgcn_readlane(lv, 0), lv1 = __builtin_amdgcn_readlane(lv, 16),
                              lv2 = __builtin_amdgcn_readlane(lv, 32), lv3 = __builtin_amdgcn_readlane(lv, 48);
                    const int lvm = max(max(lv0, lv1), max(lv2, lv3));
                    float m = -INFINITY;
                    for (int r0 = 0; r0 < lvm; r0 += 16) {
                        const int r = r0 + sub;
                        if (r < lv) {
                            const int row = off + r;
                            const float sc = ((sp[row] + sp[row + PR]) + (sp[row + 2 * PR] + sp[row + 3 * PR])) * inv_temp + s_mb[row];
                            prow[row] = sc;
                            m = fmaxf(m, sc);
                        }
                    }
                    m = row16_max(m);
                    float z = 0.f;
                    for (int r0 = 0; r0 < lvm; r0 += 16) {
                        const int r = r0 + sub;
                        if (r < lv) {
                            const float sc = prow[off + r];
                            const float e = (sc != -INFINITY) ? __expf(sc - m) : 0.f;
                            prow[off + r] = e;
                            z += e;
                        }
                    }
                    z = row16_sum(z);
                    const float rz = 1.0f / z;          // no live row: 0 * inf = NaN below, like the reference's softmax of -inf
                    const float dead = lv == 0 ? rz * 0.f : 0.f;
                    const int l8 = lv <= 8 ? 8 : (lv + 7) & ~7;
                    const int l8m = lvm <= 8 ? 8 : (lvm + 7) & ~7;
                    for (int r0 = 0; r0 < l8m; r0 += 16) {
                        const int r = r0 + sub;
                        if (act && r < l8) prow[off + r] = r < lv ? prow[off + r] * rz : dead;
                    }
                    if (attn && act) {
                        float* arow = attn + ((size_t)h * B + first + sl) * L;
                        for (int r = sub; r < L; r += 16) arow[r] = r < lv ? prow[off + r] : dead;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(s_smdone + n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
            const int n = t - hl, h = h0 + n;
            kv_gemm<NT, true, false>(acc, f, a_lo, a_hi, wsr, wb, wb_next, 0u, [](int) { return 0; });
            // ---- weighted sums per 8-row block (samples are 8-row aligned): registers 4g..4g+3 of both halves are the block's
            //      rows; one swap + add per pair of blocks, 128-B rows of block sums to LDS
            lds_wait_ge(s_smdone + n, 1);
            const float* pp = s_p + n * PR + 4 * half;
            float* blk = s_blk + (n * 4 + wq) * (PR / 8) * 32;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                float bs[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 pv = *reinterpret_cast<const f32x4*>(pp + 32 * i + 8 * g);
                    bs[g] = fmaf(pv[3], acc[i][4 * g + 3], fmaf(pv[2], acc[i][4 * g + 2], fmaf(pv[1], acc[i][4 * g + 1], pv[0] * acc[i][4 * g])));
                }
                blk[(4 * i) * 32 + lane] = halves_sum2(bs[0], bs[1]);          // lanes 0-31: block 4i, lanes 32-63: block 4i + 1
                blk[(4 * i + 2) * 32 + lane] = halves_sum2(bs[2], bs[3]);
            }
            // ---- a sample's blocks added up, two samples at a time (one per half wave), lane = head dim
            const int d = lane & 31;
            const float bias = bv ? bv[h * DK + wq * 32 + d] : 0.f;
            for (int s0 = 0; s0 < cnt; s0 += 2) {
                const int sl = s0 + half;
                if (sl < cnt) {
                    const int lv = s_slen[sl];
                    const int k0 = s_soff[sl] >> 3, nb = lv <= 8 ? 1 : (lv + 7) >> 3;
                    float acc1 = 0.f;
                    for (int k = 0; k < nb; ++k) acc1 += blk[(k0 + k) * 32 + d];
                    o[((size_t)(first + sl) * H + h) * DK + wq * 32 + d] = acc1 + bias;
                }
            }
        }
        t = t_next;
    }
}

__global__ __launch_bounds__(NTHR) void sq_mha32_packed_kernel(const float* __restrict__ qh, const unsigned short* __restrict__ bank,
                                                               const float* __restrict__ mask, const int* __restrict__ plan,
                                                               int B, int L, int H, const unsigned short* __restrict__ Wp,
                                                               const float* __restrict__ bv, float temp, float* __restrict__ o,
                                                               float* __restrict__ attn, int* __restrict__ status) {
    // a plan of the other kind (the grouped split-bf16 core's has the same size) or of another batch: nothing is trusted, nothing
    // is written, the library's status word says so (uniform over the launch: every workgroup leaves here)
    if (plan[2] != mg_plan::kind_word(8, PR, PS) || plan[1] != B) {
        if (threadIdx.x == 0 && status) __hip_atomic_store(status, MGNNS_STATUS_BAD_PLAN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_mb = reinterpret_cast<float*>(smem + P_OFF_MB);
    int* s_src = reinterpret_cast<int*>(smem + P_OFF_SRC);
    int* s_rsl = reinterpret_cast<int*>(smem + P_OFF_RSL);
    int* s_int = reinterpret_cast<int*>(smem + P_OFF_INT);
    float* s_q = reinterpret_cast<float*>(smem + P_OFF_Q);
    const int tid = threadIdx.x;
    const int ng = plan[0];
    const int h0 = (int)blockIdx.y * PHL;
    const int hl = H - h0 < PHL ? H - h0 : PHL;
    const uint4* xb = reinterpret_cast<const uint4*>(bank);
    for (int g = blockIdx.x; g < ng; g += gridDim.x) {
        const int first = plan[PLAN_HDR + 4 * g];
        int cnt = plan[PLAN_HDR + 4 * g + 1], rows = plan[PLAN_HDR + 4 * g + 2];
        cnt = cnt > PS ? PS : cnt;                       // (what the LDS maps hold, whatever the plan says)
        rows = rows > PR ? PR : rows;
        __syncthreads();                                 // the previous group's LDS is free
        if (tid < 8) s_int[tid] = 0;                     // tickets, arrival counts
        if (tid < PS) {
            s_int[8 + tid] = tid < cnt ? plan[PLAN_HDR + 4 * B + 2 * (first + tid)] : 0;
            s_int[24 + tid] = tid < cnt ? plan[PLAN_HDR + 4 * B + 2 * (first + tid) + 1] : 0;
        }
        __syncthreads();
        const int nt = (rows + RT - 1) / RT;
        const int n_sel = nt <= 1 ? 1 : nt <= 2 ? 2 : 4;
        if (tid < PR) {                                   // the row map: sample, source row, mask bias
            int sl = 0, src = -1;
            float mb = -INFINITY;
            if (tid < rows) {
                for (int s = 1; s < cnt; ++s) sl = tid >= s_int[8 + s] ? s : sl;
                const int pos = tid - s_int[8 + sl], lv = s_int[24 + sl];
                if (pos < lv) {
                    src = (first + sl) * L + pos;
                    mb = mask[(size_t)src] != 0.0f ? 0.0f : -INFINITY;
                }
            }
            s_rsl[tid] = sl;
            s_src[tid] = src;
            s_mb[tid] = mb;
        }
        for (int i = tid; i < cnt * hl * (DK / 4); i += NTHR) {       // the group's projected queries of this workgroup's heads
            const int sl = i / (hl * (DK / 4)), r = i - sl * (hl * (DK / 4));
            *reinterpret_cast<f32x4*>(s_q + sl * PQS + r * 4) =
                *reinterpret_cast<const f32x4*>(qh + ((size_t)(first + sl) * H + h0) * DK + r * 4);
        }
        __syncthreads();
        // rows of the class beyond the group's rows, and the alignment rows of a sample, read zeros (bank row 0's padding)
        stage_rows(smem, n_sel * RT, xb + (CH - 1), [&](int row) {
            const int src = row < PR ? s_src[row] : -1;
            return src >= 0 ? xb + (size_t)src * CH : (const uint4*)nullptr;
        });
        switch (n_sel) {
            case 1: packed_body<1>(smem, B, L, H, h0, hl, first, cnt, Wp, bv, temp, o, attn); break;
            case 2: packed_body<2>(smem, B, L, H, h0, hl, first, cnt, Wp, bv, temp, o, attn); break;
            default: packed_body<4>(smem, B, L, H, h0, hl, first, cnt, Wp, bv, temp, o, attn); break;
        }
    }
}

}  // namespace

#ifdef MG_MHA32_TRACE
extern "C" int mgnns_debug_mha32_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mha32_trace), sizeof(unsigned long long) * 4 * 64) == hipSuccess ? 0 : 1;
}
#endif

extern "C" size_t mgnns_sq_mha32_packed_weight_bytes(int H) { return (size_t)H * 2 * 4 * KS * 64 * 16; }

extern "C" int mgnns_sq_mha32_pack_weights_bf16(const float* Wk, const float* Wv, int H, int dk, int D, void* Wp,
                                                mgnns_stream_t stream) {
    MG_REQUIRE(Wk && Wv && Wp, "mgnns_sq_mha32_pack_weights_bf16: null pointer");
    MG_REQUIRE(dk == DK && H > 0 && D > 0 && D <= KS * 16, "mgnns_sq_mha32_pack_weights_bf16: unsupported dk=%d D=%d", dk, D);
    const size_t total = (size_t)H * 2 * 4 * KS * 64;
    hipLaunchKernelGGL(pack_kv_weights32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Wk, Wv,
                       H, D, reinterpret_cast<unsigned short*>(Wp));
    MG_CHECK_LAUNCH("mgnns_sq_mha32_pack_weights_bf16");
    return 0;
}

extern "C" size_t mgnns_sq_mha32_plan_ints(int B) { return (size_t)PLAN_HDR + 6 * (size_t)(B > 0 ? B : 0); }

extern "C" int mgnns_sq_mha32_plan(const float* mask, int B, int L, int32_t* plan, mgnns_stream_t stream) {
    MG_REQUIRE(mask && plan, "mgnns_sq_mha32_plan: null pointer");
    MG_REQUIRE(B >= 0 && B <= 4096 && L > 0 && L <= PR, "mgnns_sq_mha32_plan: B=%d (<= 4096), L=%d (1..%d) unsupported", B, L, PR);
    const size_t lds = mg_plan::lds_bytes(B);
    MG_DYN_LDS(sq_mha32_plan_kernel, lds);
    hipLaunchKernelGGL(sq_mha32_plan_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, mask, B, L, plan);
    MG_CHECK_LAUNCH("mgnns_sq_mha32_plan");
    return 0;
}

extern "C" int mgnns_sq_mha32_core_bf16_fwd(const float* qh, const void* bank_bf16, const float* mask, int B, int L, int ld,
                                            int H, int dk, const void* Wp, const float* bk, const float* bv, float* o,
                                            float* attn, const int32_t* plan, mgnns_stream_t stream) {
    (void)bk;                           // q.b_k shifts every score of a head equally: softmax invariant
    MG_REQUIRE(qh && bank_bf16 && Wp && o, "mgnns_sq_mha32_core_bf16_fwd: null pointer");
    MG_REQUIRE(dk == DK, "mgnns_sq_mha32_core_bf16_fwd: d_kv=%d unsupported (128 only)", dk);
    MG_REQUIRE(ld == CH * 8, "mgnns_sq_mha32_core_bf16_fwd: bank row length %d must be %d (bf16, zero padded)", ld, CH * 8);
    MG_REQUIRE(B >= 0 && H > 0 && L > 0 && L <= LMAX, "mgnns_sq_mha32_core_bf16_fwd: L=%d unsupported (1..%d)", L, LMAX);
    MG_REQUIRE((double)H * B * L * 4 < 2147483648.0, "mgnns_sq_mha32_core_bf16_fwd: attn output beyond 2 GiB (B=%d)", B);
    MG_REQUIRE(H * DK <= QMAX, "mgnns_sq_mha32_core_bf16_fwd: n_head=%d unsupported (<= %d)", H, QMAX / DK);
    MG_REQUIRE(mg_aligned16(bank_bf16) && mg_aligned16(Wp) && mg_aligned16(qh), "mgnns_sq_mha32_core_bf16_fwd: qh/bank/Wp must be 16-byte aligned");
    MG_REQUIRE(!plan || (mask && L <= PR), "mgnns_sq_mha32_core_bf16_fwd: a packing plan needs a mask and L <= %d", PR);
    if (B == 0) return 0;
    const int n_cu = mg_cu_count();
    if (n_cu <= 0) return MGNNS_ERR_LAUNCH;
    const float temp = (float)sqrt((double)dk);
    const int pairs = (H + 1) / 2;
    if (plan) {
        MG_DYN_LDS(sq_mha32_packed_kernel, P_SMEM_BYTES);
        // groups are device data: the grid is sized for ~5 samples per group (MVSA-like lengths pack 6 per group; a workgroup
        // walks groups g, g + gx, ... if there are more) -- NOT for the worst case: an idle workgroup of this launch still needs
        // a CU with 120 KB of free LDS to be scheduled at all, and on a busy chip every one of them queues behind somebody
        // else's workgroup before the launch can retire
        int gx = (B + 4) / 5;
        if (gx > n_cu) gx = n_cu;
        hipLaunchKernelGGL(sq_mha32_packed_kernel, dim3(gx, pairs), dim3(NTHR), P_SMEM_BYTES, (hipStream_t)stream, qh,
                           reinterpret_cast<const unsigned short*>(bank_bf16), mask, plan, B, L, H,
                           reinterpret_cast<const unsigned short*>(Wp), bv, temp, o, attn, mg_status_word());
        MG_CHECK_LAUNCH("mgnns_sq_mha32_core_bf16_fwd (packed)");
        return 0;
    }
    MG_DYN_LDS(sq_mha32_core_kernel, SMEM_BYTES);
    int gy = 1;                         // one workgroup per sample owns all head pairs when the batch fills the chip
    while (gy < pairs && B * gy < n_cu) gy *= 2;
    if (const int e = mg_env_int("MGNNS_MHA_SPLIT", 0, 4)) gy = e;        // measurement knob: workgroups per sample
    if (gy > pairs) gy = pairs;
    hipLaunchKernelGGL(sq_mha32_core_kernel, dim3(B, gy), dim3(NTHR), SMEM_BYTES, (hipStream_t)stream, qh,
                       reinterpret_cast<const unsigned short*>(bank_bf16), mask, B, L, H,
                       reinterpret_cast<const unsigned short*>(Wp), bv, temp, o, attn);
    MG_CHECK_LAUNCH("mgnns_sq_mha32_core_bf16_fwd");
    return 0;
}
