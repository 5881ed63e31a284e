// bf16-operand variant of the fused single-query multi-head attention (submodules.py:55-119, len_q == 1):
// K/V projections on v_mfma_f32_16x16x32_bf16 with fp32 accumulation; scores, mask, softmax and the
// probability-weighted sum stay fp32.  K and V never leave the accumulators.
//
// Geometry (MI355X): one 512-thread workgroup per sample walks the heads in PAIRS.  The sample's memory
// bank X [L<=208, 320] (bf16, 640-B rows) is staged ONCE into LDS with a 672-B row stride (42 x 16 B:
// conflict-free for the ds_read_b128 A-fragment pattern row = lane&15, chunk = lane>>4) and reused by
// every head.  Waves 0-3 own head h0, waves 4-7 head h1; each wave owns 32 head dims (2 column tiles) x
// up to 13 row tiles -> 26 accumulator tiles.  The B operand (W_k,h / W_v,h) is read straight from global
// memory in a pre-packed fragment-major layout (1 KiB contiguous per fragment, L2 resident), so LDS only
// carries X.  Per head pair: phase 0 = K tiles -> scores -> softmax, phase 1 = V tiles -> weighted sum.
// Row tiles behind the last unmasked position are skipped (their probability is exactly 0).
#include "common.hpp"
#include "mha_tail_body.hpp"      // (defines bf16x8)

#ifndef MG_MHA_ABLATE
#define MG_MHA_ABLATE 0      // measurement builds only (tools/dev/build_variant.py): 1 = no weight-fragment reloads, 2 = no bank-fragment reloads, 4 = the bank is not staged at all, 8 = a 12-tile class (L <= 192)
#endif
#ifdef MG_MHA_TRACE
// profiling aid (off by default): s_memtime stamps of wave 0 / wave 4 of two workgroups at every phase boundary
__device__ unsigned long long g_mha_trace[4][64];
#define MG_STAMP(slot)                                                                              \
    do {                                                                                            \
        if ((threadIdx.x & 255) == 0 && (blockIdx.x == 0 || blockIdx.x == 129) && blockIdx.y == 0)  \
            g_mha_trace[(blockIdx.x ? 2 : 0) + (threadIdx.x >> 8)][(slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define MG_STAMP(slot) do { } while (0)
#endif

namespace {

constexpr int MT = 13;                  // row tiles of 16 (L <= 208)
constexpr int LMAX = MT * 16;
constexpr int KP = 320;                 // model dim padded to 10 k-steps of 32
constexpr int KSTEPS = KP / 32;
constexpr int CH = KP / 8;              // 40 16-byte chunks per row
constexpr int LSTR = 42;                // LDS row stride in chunks (672 B)
constexpr int DK = 128;
#ifndef MG_MHA_NTHR
#define MG_MHA_NTHR 512          // (measurement: 768 = three waves per SIMD at <= 168 registers)
#endif
constexpr int NTHR = MG_MHA_NTHR;
constexpr int QMAX = 2048;              // floats of the projected query kept in LDS (H * 128 <= QMAX)

__device__ __forceinline__ unsigned short f2bf(float x) {      // round-to-nearest-even
    unsigned int u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// Wp[h][kv][nt][ks][lane][8] = W_kv[h*128 + nt*16 + (lane&15)][ks*32 + (lane>>4)*8 + j]  (0 beyond D)
__global__ __launch_bounds__(256) void pack_kv_weights_kernel(const float* __restrict__ Wk, const float* __restrict__ Wv,
                                                              int H, int D, unsigned short* __restrict__ Wp) {
    const size_t total = (size_t)H * 2 * 8 * KSTEPS * 64;       // fragments-lanes
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        size_t r = i >> 6;
        const int ks = (int)(r % KSTEPS); r /= KSTEPS;
        const int nt = (int)(r & 7); r >>= 3;
        const int kv = (int)(r & 1);
        const int h = (int)(r >> 1);
        const float* W = kv ? Wv : Wk;
        const int row = h * DK + nt * 16 + (lane & 15);
        const int k0 = ks * 32 + (lane >> 4) * 8;
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

// y[r, 0:ld] = bf16(x[r, 0:D]) zero padded to ld
__global__ __launch_bounds__(256) void cast_pad_bf16_kernel(const float* __restrict__ x, size_t rows, int D, int ld,
                                                            unsigned short* __restrict__ y) {
    const size_t total = rows * (size_t)(ld / 8);
    const int c8n = ld / 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / c8n;
        const int c0 = (int)(i - r * c8n) * 8;
        unsigned short v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (c0 + j < D) ? f2bf(x[r * D + c0 + j]) : (unsigned short)0;
        uint4 o;
        o.x = v[0] | ((unsigned)v[1] << 16);
        o.y = v[2] | ((unsigned)v[3] << 16);
        o.z = v[4] | ((unsigned)v[5] << 16);
        o.w = v[6] | ((unsigned)v[7] << 16);
        reinterpret_cast<uint4*>(y)[i] = o;
    }
}

// Sum over the four 16-lane rows of the wave for FOUR values at once (a transposing reduction): on return the rows of the
// result hold the row sums of [a, c, b, d] -- row 0: a, row 1: c, row 2: b, row 3: d.  v_permlane32_swap exchanges the upper
// half of its first operand with the lower half of its second, v_permlane16_swap the odd rows of the first with the even rows
// of the second, so one swap + one add folds TWO values by one level: 3 swaps + 3 adds for four tiles (the one-value form, both
// operands the same register, cost 2 swaps + 2 adds per tile).  Inline asm: both registers of a swap are read AND written (hipcc
// 7.2's builtin loses the second result here); the s_nop 1 on either side cover the VALU-write -> swap-read and swap-write ->
// VALU-read hazards, which the compiler's hazard recogniser does not see through an asm block.
__device__ __forceinline__ float rows4_sum4(float a, float b, float c, float d) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));      // a = [a.lo, b.lo], b = [a.hi, b.hi] (same for c, d)
    float ab = a + b, cd = c + d;                             // halves: [a: r0+r2, r1+r3 | b: r0+r2, r1+r3]
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(ab), "+v"(cd));
    return ab + cd;                                           // rows: [a, c, b, d]
}

// Cross-wave hand-over inside the workgroup goes through LDS counters, not s_barrier (see mha_body).  LDS operations of a wave
// complete in order: lgkmcnt(0) in front of an arrival publishes this wave's LDS writes to whoever sees the count.
__device__ __forceinline__ int lds_arrive(int* ctr, int lane) {          // -> the count before this arrival (wave-uniform)
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

// LDS map behind the staged bank
constexpr int NSLOT_P = 8;                                               // probability rows kept (heads between softmax and weighted sum)
constexpr int MAXH = QMAX / DK;                                          // heads a workgroup can own
constexpr size_t OFF_PART = (size_t)LMAX * LSTR * 16;                    // float [2][4][LMAX] partial scores (head parity, slice)
constexpr size_t OFF_P = OFF_PART + 2 * 4 * LMAX * sizeof(float);       // float [NSLOT_P][LMAX] probabilities
constexpr size_t OFF_MB = OFF_P + NSLOT_P * LMAX * sizeof(float);       // float [LMAX] mask bias: 0 or -inf
constexpr size_t OFF_INT = OFF_MB + LMAX * sizeof(float);               // int [16 + 3 * MAXH]: live rows, tickets, arrival counts
constexpr size_t OFF_Q = OFF_INT + (16 + 3 * MAXH) * sizeof(int);       // float [QMAX] this sample's projected query
constexpr size_t SMEM_BYTES = OFF_Q + QMAX * sizeof(float);
static_assert(SMEM_BYTES <= 160 * 1024, "LDS");
static_assert(OFF_Q % 16 == 0 && OFF_P % 16 == 0, "LDS alignment");

// Weight fragments in flight ACROSS a unit boundary: the first BD k-steps' B fragments of the NEXT unit's weight stream
// are requested at the tail of the current GEMM, so the epilogue runs with the next unit's first operands already on their
// way instead of exposing an L2 round trip at every unit start.
template <int NMT>
struct Frags {
    static constexpr int BD = NMT <= 4 ? 5 : 2;      // must divide KSTEPS (ring slots carry over units)
#ifndef MG_MHA_RING
#define MG_MHA_RING 8
#endif
    // bank fragments in flight inside a GEMM: a ring over the (k-step, row tile) sequence -- a fragment is requested RA tiles
    // (2 RA MFMAs) ahead of its use; the 13-tile class (104 accumulators) has room for MG_MHA_RING of them
    static constexpr int TOTAL = KSTEPS * NMT;
    static constexpr int RA = TOTAL < MG_MHA_RING ? TOTAL : MG_MHA_RING;
    uint4 bq[BD][2];
};

// The packed weights are read through a buffer resource: one VGPR (lane * 16) addresses every fragment, the
// fragment itself is selected by a wave-uniform byte offset in an SGPR.  (64-bit per-lane pointers for the two live
// weight streams cost the 13-tile class enough registers to spill its head-pair loop state.)
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct WStream {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;                                   // lane * 16
};
__device__ __forceinline__ uint4 wfrag(const WStream& w, int soff) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(w.rsrc, w.voff, soff, 0));
}
constexpr int FRAG = 1024;                      // bytes per fragment
template <int NMT>
__device__ __forceinline__ void frags_prime_b(Frags<NMT>& f, const WStream& w, int wb) {
#pragma unroll
    for (int d = 0; d < Frags<NMT>::BD; ++d) {
        f.bq[d][0] = wfrag(w, wb + d * FRAG);
        f.bq[d][1] = wfrag(w, wb + (KSTEPS + d) * FRAG);
    }
}
// compile-time loop: f(IC<0>{}), f(IC<1>{}), ... -- the index is a constant expression inside f (immediate offsets / counts of
// inline-asm instructions need one)
template <int N> struct IC { static constexpr int v = N; };
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(IC<I>{});
        static_for<I + 1, N>(f);
    }
}

// acc[i][j] += X[tile i] * W^T[tile j] over the padded model dim, for a COMPILE-TIME number of live row tiles.
// The kernel is bound by instruction ISSUE, not by the matrix pipe alone: a SIMD issues ~1.3 other instructions in the shadow
// of one 16-cycle MFMA (tools/dev/micro/mfma_valu.hip), and the compiler-scheduled form of this loop carried 2.2 per MFMA --
// an s_waitcnt in front of every row tile, a v_add per LDS address beyond the 64-KiB immediate range.  So the loop is written
// out: bank fragments are read with inline-asm ds_read_b128 from TWO base registers with immediate offsets (rows 0-111 /
// 112-207), into a ring that runs RA tiles ahead of the MFMAs, and ONE hand-counted s_waitcnt lgkmcnt per pair of row tiles
// (LDS operations of a wave complete in order; an extra younger LDS operation in flight only makes a counted wait stricter).
// Two MFMAs consume a fragment and the ds_read that refills its ring slot follows them at once: one LDS read per two MFMAs.
// B fragments (global, fragment-major, L2 resident, compiler-visible loads) run BD k-steps ahead; the last k-steps prefetch
// the NEXT unit's first fragments (next_stream() draws that unit).  sched_barrier(0) fences pin the order: nothing ties the
// MFMAs to the asm reads' waits except that no instruction is scheduled across a fence.
template <int NMT, typename NextStream>
__device__ __forceinline__ void kv_gemm(f32x4 (&acc)[MT][2], Frags<NMT>& f, unsigned a_lo, unsigned a_hi,
                                        const WStream& w, int wb, NextStream&& next_stream, int trace_base = -1) {
    constexpr int BD = Frags<NMT>::BD, RA = Frags<NMT>::RA, TOTAL = Frags<NMT>::TOTAL;
    constexpr int HOOK_KS = KSTEPS - BD - 1 > 0 ? KSTEPS - BD - 1 : 0;      // k-step in front of which the next unit is drawn
    constexpr int SPLIT = 7;                            // row tiles 0..6 from a_lo, 7..12 from a_hi = a_lo + 7 * 16 rows
#ifndef MG_MHA_GROUP
#define MG_MHA_GROUP 2
#endif
    constexpr int GRP = MG_MHA_GROUP < RA ? MG_MHA_GROUP : 1;       // row tiles per counted wait
    u32x4 ga[RA];
    int wb_next = 0;
    // fragment n of the (k-step, row tile) sequence -> ring slot n % RA
    auto fetch = [&](auto nc) {
        constexpr int n = decltype(nc)::v;
        constexpr int ks = n / NMT, i = n % NMT;
        if constexpr (i < SPLIT) ga[n % RA] = mg_lds_read128<(i * 16 * LSTR + ks * 4) * 16>(a_lo);
        else ga[n % RA] = mg_lds_read128<((i - SPLIT) * 16 * LSTR + ks * 4) * 16>(a_hi);
    };
    static_for<0, RA>(fetch);
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, KSTEPS>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::v;
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, f.bq[ks % BD][0]);
        const bf16x8 b1 = __builtin_bit_cast(bf16x8, f.bq[ks % BD][1]);
#ifdef MG_MHA_TRACE
        if (trace_base >= 0) MG_STAMP(trace_base + ks);
#endif
        if (ks == HOOK_KS) wb_next = next_stream();
        static_for<0, (NMT + GRP - 1) / GRP>([&](auto gc) {
            constexpr int i0 = decltype(gc)::v * GRP;
            constexpr int cnt = i0 + GRP <= NMT ? GRP : NMT - i0;      // row tiles of this group
            constexpr int n0 = ks * NMT + i0;
            // reads issued so far: fragments 0 .. min(n0 + RA, TOTAL) - 1; fragments n0 .. n0 + cnt - 1 must have landed
            constexpr int issued = n0 + RA < TOTAL ? n0 + RA : TOTAL;
            mg_lds_wait<issued - n0 - cnt>();
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, cnt>([&](auto jc) {
                constexpr int i = i0 + decltype(jc)::v;
                const bf16x8 av = __builtin_bit_cast(bf16x8, ga[(ks * NMT + i) % RA]);
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, av, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, av, acc[i][1], 0, 0, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
#if !(MG_MHA_ABLATE & 2)
            static_for<0, cnt>([&](auto jc) {
                constexpr int n2 = n0 + decltype(jc)::v + RA;
                if constexpr (n2 < TOTAL) fetch(IC<n2>{});
            });
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
#if !(MG_MHA_ABLATE & 1)
        if (ks + BD < KSTEPS) {
            f.bq[ks % BD][0] = wfrag(w, wb + (ks + BD) * FRAG);
            f.bq[ks % BD][1] = wfrag(w, wb + (KSTEPS + ks + BD) * FRAG);
        } else {                                     // next unit's k-steps 0..BD-1
            f.bq[ks % BD][0] = wfrag(w, wb_next + (ks + BD - KSTEPS) * FRAG);
            f.bq[ks % BD][1] = wfrag(w, wb_next + (KSTEPS + ks + BD - KSTEPS) * FRAG);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
    });
}

// Everything after the bank is staged, for a compile-time tile-count class.
//
// Work is cut into UNITS = (head, K or V projection) x (slice of 32 head dims).  The two waves of a slice (w and w + 4: they
// share a SIMD, hence its matrix pipe) draw the slice's units from an LDS ticket counter in the order K(h0) K(h1) V(h0) V(h1)
// K(h2) ..., so whichever wave is free takes the next one: while one wave is in a VALU-only epilogue (scores, weighted sum:
// ~1.5 k cycles) its partner's MFMAs keep the pipe busy, and the two finish within one unit of each other however the
// hardware arbitrates between them (rounds 1-2: fixed heads per half-workgroup in lock step through five s_barriers per head
// pair: the pipe idled through every epilogue; per-half barriers alone let the older half run away and finish 14 k cycles early).
// No s_barrier after the staging one; hand-over through LDS counters:
//   K unit   GEMM -> this slice's partial scores of the head -> arrival at the head's count; the LAST of the four slices to
//            arrive runs the head's softmax (one wave, four positions per lane) and publishes the probabilities
//   V unit   GEMM -> wait for the head's probabilities (published a whole GEMM earlier, as a rule) -> weighted sum -> o
// Guards make the ring slots safe under ANY progress order: partial scores of local head n reuse the slot of head n - 2 (wait
// for its softmax), probabilities of head n the row of head n - 8 (wait for its four weighted sums).  A unit never waits for a
// later ticket, so the queue cannot deadlock.
// COH: `o` is stored with system-scope write-through stores (aux sc0 | sc1) because ANOTHER workgroup of this launch reads it
// (the fused layer kernel below); otherwise ordinary stores
template <int NMT, bool COH>
__device__ __forceinline__ void mha_body(unsigned char* smem, int B, int L, int H, const unsigned short* __restrict__ Wp,
                                         const float* __restrict__ bv, float temp, float* __restrict__ o,
                                         float* __restrict__ attn, int lvalid) {
    uint4* Xs = reinterpret_cast<uint4*>(smem);
    float* s_part = reinterpret_cast<float*>(smem + OFF_PART);
    float* s_p = reinterpret_cast<float*>(smem + OFF_P);
    const float* s_mb = reinterpret_cast<const float*>(smem + OFF_MB);
    int* s_int = reinterpret_cast<int*>(smem + OFF_INT);
    int* s_ticket = s_int + 4;                          // [4] next unit of a slice
    int* s_kdone = s_int + 16;                          // [MAXH] slices that delivered their partial scores of local head n
    int* s_smdone = s_int + 16 + MAXH;                  // [MAXH] 1 = probabilities of local head n published
    int* s_pvdone = s_int + 16 + 2 * MAXH;              // [MAXH] slices done with the probabilities of local head n
    const float* s_q = reinterpret_cast<const float*>(smem + OFF_Q);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform values live in SGPRs
    const int wq = wave & 3;                                        // slice: head dims 32 wq ... 32 wq + 31
    const int b = blockIdx.x;
    // LDS byte address of this lane's 16 bytes of row tile 0 / 7, k-step 0 (+ i*16*LSTR*16 + ks*64 as immediates)
    const unsigned a_lo = mg_lds_addr(Xs + (lane & 15) * LSTR + (lane >> 4)), a_hi = a_lo + 7 * 16 * LSTR * 16;
    const float inv_temp = 1.0f / temp;
    WStream wsr;
    wsr.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Wp), 0, 0x7fffffff, 0x00027000);
    wsr.voff = lane * 16;
    // this workgroup's head pairs: blockIdx.y, + gridDim.y, ...; ticket t -> pair t / 4, head t & 1 of the pair, V if t & 2
    const int pairs = (H + 1) / 2;
    const int npairs = (pairs - (int)blockIdx.y + (int)gridDim.y - 1) / (int)gridDim.y;
    const int nunits = npairs * 4;
    auto head_of = [&](int t) { return ((int)blockIdx.y + (t >> 2) * (int)gridDim.y) * 2 + (t & 1); };
    auto draw = [&]() {                                 // next unit of this slice; tickets of a head beyond H (odd H) are skipped
        int t;
        do {
            int v = 0;
            if (lane == 0) v = __hip_atomic_fetch_add(s_ticket + wq, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            t = __builtin_amdgcn_readfirstlane(v);
        } while (t < nunits && head_of(t) >= H);
        return t;
    };
    // weight stream of a unit for this wave (byte offset of its first fragment, wave-uniform); past the last unit: a harmless re-read
    auto wstream = [&](int t) {
        const int h = t < nunits ? head_of(t) : 0;
        return (((h * 2 + ((t >> 1) & 1)) * 8 + wq * 2) * KSTEPS) * FRAG;
    };
    // o row of this sample / b_v: buffer resources (uniform base, 32-bit per-lane offsets)
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(o + (size_t)b * H * DK, 0, H * DK * 4, 0x00027000);
    const __amdgpu_buffer_rsrc_t bv_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bv), 0, bv ? H * DK * 4 : 0, 0x00027000);
    const __amdgpu_buffer_rsrc_t attn_rsrc = __builtin_amdgcn_make_buffer_rsrc(attn, 0, attn ? 0x7fffffff : 0, 0x00027000);
    // the tiles are computed TRANSPOSED (rows = head dims, columns = bank rows): this lane's accumulator element
    // [i][j][r] is head dim d(j,r) = wq*32 + 16j + 4*(lane>>4) + r of bank row 16i + (lane&15)
    const int dbase = wq * 32 + (lane >> 4) * 4;

    int t = draw();
    Frags<NMT> f;
    frags_prime_b<NMT>(f, wsr, wstream(t));             // weight fragments on their way while the bank DMA lands
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's LDS-DMA pieces of the bank (not tracked by hipcc)
    __syncthreads();                                    // ... and every other wave's
    MG_STAMP(2);
    int stamp = 3;
    (void)stamp;

    while (t < nunits) {
        const int n = (t >> 2) * 2 + (t & 1);           // workgroup-local index of the head
        const int h = head_of(t);
        const bool vunit = (t & 2) != 0;
        int t_next = nunits;
        f32x4 acc[MT][2];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        kv_gemm<NMT>(acc, f, a_lo, a_hi, wsr, wstream(t), [&]() { t_next = draw(); return wstream(t_next); });
        MG_STAMP(stamp++);
#ifdef MG_MHA_EPI_PRIO
        __builtin_amdgcn_s_setprio(MG_MHA_EPI_PRIO);
#endif

        if (!vunit) {
            // ---- partial scores of this slice's 32 head dims: in-register over the 8 dims of the lane, then across
            //      the four 16-lane groups, four row tiles per transposing reduction (rows4_sum4); the result register
            //      holds tiles [4g, 4g+2, 4g+1, 4g+3] in its rows: one 64-lane LDS write per four tiles.
            // Biases: q.(K_l + b_k) = q.K_l + q.b_k shifts every score of the head by the same constant, which the
            // softmax cancels exactly, so b_k never enters; sum_l p_l (V_l + b_v) = sum_l p_l V_l + b_v because the
            // probabilities sum to 1, so b_v is added once to the 8 outputs of the lane.
            const float* qv = s_q + h * DK;             // the query row sits in LDS since the prologue
            const f32x4 qd0 = *reinterpret_cast<const f32x4*>(qv + dbase), qd1 = *reinterpret_cast<const f32x4*>(qv + dbase + 16);
            float v[(NMT + 3) / 4 * 4];
#pragma unroll
            for (int i = 0; i < (NMT + 3) / 4 * 4; ++i) {
                float a = 0.f, c = 0.f;
                if (i < NMT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        a = fmaf(qd0[r], acc[i][0][r], a);
                        c = fmaf(qd1[r], acc[i][1][r], c);
                    }
                }
                v[i] = a + c;
            }
            if (n >= 2) lds_wait_ge(s_smdone + n - 2, 1);           // the slot's previous head has been consumed
            float* part = s_part + ((n & 1) * 4 + wq) * LMAX;
            const int rt = ((lane >> 4) & 1) * 2 + (lane >> 5);      // row tile (inside a group of 4) of this lane's row
#pragma unroll
            for (int g = 0; g < (NMT + 3) / 4; ++g) {
                const float s4 = rows4_sum4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
                if (4 * g + 3 < NMT || 4 * g + rt < NMT) part[(4 * g + rt) * 16 + (lane & 15)] = s4;
            }
            if (lds_arrive(s_kdone + n, lane) == 3) {
                // ---- last slice of the head: masked softmax over all positions, four per lane (lane, +64, +128, +192)
                if (n >= NSLOT_P) lds_wait_ge(s_pvdone + n - NSLOT_P, 4);
                const float* sp = s_part + (n & 1) * 4 * LMAX + lane;
                float sc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pos = lane + 64 * j;
                    sc[j] = -INFINITY;
                    if (pos < lvalid)
                        sc[j] = ((sp[64 * j] + sp[64 * j + LMAX]) + (sp[64 * j + 2 * LMAX] + sp[64 * j + 3 * LMAX])) * inv_temp + s_mb[pos];
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
        } else {
            // ---- o[d] = sum_l p[l] * (V[l,d] + bv[d]): p is per column here, 8 dims per lane accumulate in
            //      registers over the row tiles, one 16-lane DPP sum per dim at the end
            lds_wait_ge(s_smdone + n, 1);
            const float* pp = s_p + (n & (NSLOT_P - 1)) * LMAX + (lane & 15);
            f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
#pragma unroll
            for (int i = 0; i < NMT; ++i) {
                const float p = pp[i * 16];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    t0[r] = fmaf(p, acc[i][0][r], t0[r]);
                    t1[r] = fmaf(p, acc[i][1][r], t1[r]);
                }
            }
            if (npairs * 2 > NSLOT_P) lds_arrive(s_pvdone + n, lane);       // (only then is the row ever reused)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                t0[r] = row16_sum(t0[r]);
                t1[r] = row16_sum(t1[r]);
            }
            if ((lane & 15) == 0) {
                const int hoff = h * DK * 4;            // wave-uniform byte offset of the head
                if (bv) {
                    const f32x4 vb0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bv_rsrc, dbase * 4, hoff, 0));
                    const f32x4 vb1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bv_rsrc, dbase * 4 + 64, hoff, 0));
#pragma unroll
                    for (int r = 0; r < 4; ++r) { t0[r] += vb0[r]; t1[r] += vb1[r]; }
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, t0), o_rsrc, dbase * 4, hoff, COH ? 17 : 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, t1), o_rsrc, dbase * 4 + 64, hoff, COH ? 17 : 0);
            }
        }
#ifdef MG_MHA_EPI_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        MG_STAMP(stamp++);
        t = t_next;
    }
}

template <bool COH>
__device__ __forceinline__ void mha_core_part(unsigned char* smem, const float* __restrict__ qh,
                                                                const unsigned short* __restrict__ bank,   // [B,L,KP] bf16
                                                                const float* __restrict__ mask, int B, int L, int H,
                                                                const unsigned short* __restrict__ Wp,
                                                                const float* __restrict__ bk, const float* __restrict__ bv,
                                                                float temp, float* __restrict__ o, float* __restrict__ attn) {
    int* s_lvalid = reinterpret_cast<int*>(smem + OFF_INT);
    float* s_mb = reinterpret_cast<float*>(smem + OFF_MB);
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const uint4* xb = reinterpret_cast<const uint4*>(bank) + (size_t)b * L * CH;
    MG_STAMP(0);

    // ---- live rows, mask bias, the unit queue's counters ---------------------------------------------------
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
        __syncthreads();
    }
    const int lvalid = *s_lvalid;
    const int n_mt = (lvalid + 15) >> 4;
    // tile-count class the branch-free GEMM body is instantiated for (dead rows inside the class are computed on
    // zero / masked data and get probability 0)
#if MG_MHA_ABLATE & 8
    const int n_sel = n_mt <= 1 ? 1 : n_mt <= 2 ? 2 : n_mt <= 4 ? 4 : n_mt <= 7 ? 7 : n_mt <= 12 ? 12 : MT;      // (measurement: what the 13th row tile costs)
#else
    const int n_sel = n_mt <= 1 ? 1 : n_mt <= 2 ? 2 : n_mt <= 4 ? 4 : n_mt <= 7 ? 7 : MT;
#endif
    const int rows_live = n_sel * 16;

    // ---- stage X (bf16) once by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, every piece in flight at
    //      once).  A DMA instruction fills 1 KiB of CONTIGUOUS LDS (M0 base + lane*16) from per-lane addresses, so
    //      the padded [row][42-chunk] image is walked linearly: lanes that fall on the 2 pad chunks of a row are
    //      switched off, rows >= L read a zero chunk (the zero padding at the end of bank row 0).
#if !(MG_MHA_ABLATE & 4)
    {
        const int lane = tid & 63, wave = tid >> 6;
        const int total = rows_live * LSTR;
        for (int pc = wave; pc * 64 < total; pc += NTHR / 64) {
            const int g = pc * 64 + lane;
            const int row = g / LSTR, c = g - row * LSTR;
            if (g < total && c < CH) {
                const uint4* src = row < L ? xb + (size_t)row * CH + c : xb + (CH - 1);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(uintptr_t)(smem + (size_t)pc * 1024),
                                                 16, 0, 0);
            }
        }
    }
#endif
    {                                   // this sample's query row -> LDS (visible after the staging barrier in mha_body)
        float* s_q = reinterpret_cast<float*>(smem + OFF_Q);
        for (int i = tid * 4; i < H * DK; i += NTHR * 4)
            *reinterpret_cast<f32x4*>(s_q + i) = *reinterpret_cast<const f32x4*>(qh + (size_t)b * H * DK + i);
    }

    switch (n_sel) {
        case 1: mha_body<1, COH>(smem, B, L, H, Wp, bv, temp, o, attn, lvalid); break;
        case 2: mha_body<2, COH>(smem, B, L, H, Wp, bv, temp, o, attn, lvalid); break;
        case 4: mha_body<4, COH>(smem, B, L, H, Wp, bv, temp, o, attn, lvalid); break;
        case 7: mha_body<7, COH>(smem, B, L, H, Wp, bv, temp, o, attn, lvalid); break;
#if MG_MHA_ABLATE & 8
        case 12: mha_body<12, COH>(smem, B, L, H, Wp, bv, temp, o, attn, lvalid); break;
#endif
        default: mha_body<MT, COH>(smem, B, L, H, Wp, bv, temp, o, attn, lvalid); break;
    }
}


__global__ __launch_bounds__(NTHR) void sq_mha_core_bf16_kernel(const float* __restrict__ qh, const unsigned short* __restrict__ bank,
                                                                const float* __restrict__ mask, int B, int L, int H,
                                                                const unsigned short* __restrict__ Wp, const float* __restrict__ bk,
                                                                const float* __restrict__ bv, float temp, float* __restrict__ o,
                                                                float* __restrict__ attn) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    mha_core_part<false>(smem, qh, bank, mask, B, L, H, Wp, bk, bv, temp, o, attn);
}

// One MyMultiHeadAttention layer in ONE launch (moudles.py:207-230): the attention core above for every sample, and the rest
// of the layer (fc, residual, LN, FFN, residual, LN, the next layer's w_qs: mha_tail_body.hpp) for a 16-sample tile run by
// whichever of the tile's core workgroups FINISHES LAST.  As a launch of its own the tail (16 workgroups that each need a
// whole CU) queued for CUs behind the chip-filling cores of the other stacks: 40-90 us on every link of the four stack
// chains (tools/trace_timeline.py).  Hand-over without agent-scope fences (a __threadfence() per workgroup doubles this
// kernel: 62.7 -> 125 us, the L2 write-back / invalidate walks are that slow): `o` leaves through system-scope
// write-through stores (sc0 | sc1), every thread waits for its own acknowledgements (vmcnt(0)), ONE relaxed agent-scope
// atomic per workgroup counts the tile's arrivals, and the last arriver reads `o` with loads that bypass the non-coherent
// caches.  Nobody spins: the last arriver does the work, the others exit.
__global__ __launch_bounds__(NTHR) void sq_mha_layer_bf16_kernel(const float* __restrict__ qh, const unsigned short* __restrict__ bank,
                                                                 const float* __restrict__ mask, int B, int L, int H,
                                                                 const unsigned short* __restrict__ Wp, const float* __restrict__ bk,
                                                                 const float* __restrict__ bv, float temp, float* __restrict__ o,
                                                                 const float* __restrict__ q_in, mg_tail::TailW w, float eps,
                                                                 float* __restrict__ out, int HKn, float* __restrict__ qh_next,
                                                                 int* __restrict__ counters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_last;
    mha_core_part<true>(smem, qh, bank, mask, B, L, H, Wp, bk, bv, temp, o, nullptr);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's write-through stores of `o` are acknowledged
    __syncthreads();
    const int tile = (int)blockIdx.x / mg_tail::ROWS;
    if (threadIdx.x == 0) {
        const int rows = B - tile * mg_tail::ROWS < mg_tail::ROWS ? B - tile * mg_tail::ROWS : mg_tail::ROWS;
        const int old = __hip_atomic_fetch_add(&counters[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == rows * (int)gridDim.y - 1;
        if (s_last) __hip_atomic_store(&counters[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;
    mg_tail::tail_bf16_body<1, true>(smem, o, H * DK, q_in, B, w, eps, out, HKn, qh_next, tile, 0, 1);
}

}  // namespace

#ifdef MG_MHA_TRACE
extern "C" int mgnns_debug_mha_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mha_trace), sizeof(unsigned long long) * 4 * 64) == hipSuccess ? 0 : 1;
}
#endif

extern "C" size_t mgnns_sq_mha_packed_weight_bytes(int H) { return (size_t)H * 2 * 8 * KSTEPS * 64 * 16; }

extern "C" int mgnns_sq_mha_pack_weights_bf16(const float* Wk, const float* Wv, int H, int dk, int D, void* Wp,
                                              mgnns_stream_t stream) {
    MG_REQUIRE(Wk && Wv && Wp, "mgnns_sq_mha_pack_weights_bf16: null pointer");
    MG_REQUIRE(dk == DK && H > 0 && D > 0 && D <= KP, "mgnns_sq_mha_pack_weights_bf16: unsupported dk=%d D=%d", dk, D);
    const size_t total = (size_t)H * 2 * 8 * KSTEPS * 64;
    hipLaunchKernelGGL(pack_kv_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Wk,
                       Wv, H, D, reinterpret_cast<unsigned short*>(Wp));
    MG_CHECK_LAUNCH("mgnns_sq_mha_pack_weights_bf16");
    return 0;
}

extern "C" int mgnns_cast_pad_bf16(const float* x, int64_t rows, int D, int ld, void* y, mgnns_stream_t stream) {
    MG_REQUIRE(x && y, "mgnns_cast_pad_bf16: null pointer");
    MG_REQUIRE(rows >= 0 && D > 0 && ld >= D && ld % 8 == 0, "mgnns_cast_pad_bf16: bad dims D=%d ld=%d", D, ld);
    if (rows == 0) return 0;
    const size_t total = (size_t)rows * (ld / 8);
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cast_pad_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (size_t)rows, D, ld,
                       reinterpret_cast<unsigned short*>(y));
    MG_CHECK_LAUNCH("mgnns_cast_pad_bf16");
    return 0;
}

extern "C" int mgnns_sq_mha_core_bf16_fwd(const float* qh, const void* bank_bf16, const float* mask, int B, int L, int ld,
                                          int H, int dk, const void* Wp, const float* bk, const float* bv, float* o,
                                          float* attn, mgnns_stream_t stream) {
    MG_REQUIRE(qh && bank_bf16 && Wp && o, "mgnns_sq_mha_core_bf16_fwd: null pointer");
    MG_REQUIRE(dk == DK, "mgnns_sq_mha_core_bf16_fwd: d_kv=%d unsupported (128 only)", dk);
    MG_REQUIRE(ld == KP, "mgnns_sq_mha_core_bf16_fwd: bank row length %d must be %d (bf16, zero padded)", ld, KP);
    MG_REQUIRE(B >= 0 && H > 0 && L > 0 && L <= LMAX, "mgnns_sq_mha_core_bf16_fwd: L=%d unsupported (1..%d)", L, LMAX);
    MG_REQUIRE((double)H * B * L * 4 < 2147483648.0, "mgnns_sq_mha_core_bf16_fwd: attn output beyond 2 GiB (B=%d)", B);
    MG_REQUIRE(H * DK <= QMAX, "mgnns_sq_mha_core_bf16_fwd: n_head=%d unsupported (<= %d)", H, QMAX / DK);
    MG_REQUIRE(mg_aligned16(bank_bf16) && mg_aligned16(Wp), "mgnns_sq_mha_core_bf16_fwd: bank/Wp must be 16-byte aligned");
    if (B == 0) return 0;
    MG_DYN_LDS(sq_mha_core_bf16_kernel, SMEM_BYTES);
    // one workgroup per sample owns all head pairs when the batch fills the chip; small batches split the pairs
    const int pairs = (H + 1) / 2;
    int gy = 1;
    const int n_cu = mg_cu_count();
    if (n_cu <= 0) return MGNNS_ERR_LAUNCH;
    while (gy < pairs && B * gy < n_cu) gy *= 2;
    if (const int e = mg_env_int("MGNNS_MHA_SPLIT", 0, 4)) gy = e;        // measurement knob: workgroups per sample
    if (mask) { if (const int e = mg_env_int("MGNNS_MHA_SPLIT_MASKED", 0, 5)) gy = e; }
    if (gy > pairs) gy = pairs;
    const float temp = (float)sqrt((double)dk);
    hipLaunchKernelGGL(sq_mha_core_bf16_kernel, dim3(B, gy), dim3(NTHR), SMEM_BYTES, (hipStream_t)stream, qh,
                       reinterpret_cast<const unsigned short*>(bank_bf16), mask, B, L, H,
                       reinterpret_cast<const unsigned short*>(Wp), bk, bv, temp, o, attn);
    MG_CHECK_LAUNCH("mgnns_sq_mha_core_bf16_fwd");
    return 0;
}

extern "C" int mgnns_sq_mha_layer_bf16_fwd(const float* qh, const void* bank_bf16, const float* mask, int B, int L, int ld, int H, int dk,
                                           const void* Wp, const float* bk, const float* bv, float* o_scratch, const float* q_in,
                                           int d_model, const void* const* packed /* fc, w1, w2, wq_next: (hi, lo) pairs */,
                                           const float* fc_b, const float* ln1_gamma, const float* ln1_beta, const float* b1,
                                           const float* b2, const float* ln2_gamma, const float* ln2_beta, float eps, float* out,
                                           const float* bq_next, int HK_next, float* qh_next, int* tile_counters,
                                           mgnns_stream_t stream) {
    MG_REQUIRE(qh && bank_bf16 && Wp && o_scratch && q_in && packed && fc_b && ln1_gamma && ln1_beta && b1 && b2 && ln2_gamma &&
               ln2_beta && out && tile_counters, "mgnns_sq_mha_layer_bf16_fwd: null pointer");
    MG_REQUIRE(dk == DK, "mgnns_sq_mha_layer_bf16_fwd: d_kv=%d unsupported (128 only)", dk);
    MG_REQUIRE(d_model == mg_tail::D, "mgnns_sq_mha_layer_bf16_fwd: d_model=%d unsupported (300 only)", d_model);
    MG_REQUIRE(ld == KP, "mgnns_sq_mha_layer_bf16_fwd: bank row length %d must be %d (bf16, zero padded)", ld, KP);
    MG_REQUIRE(B >= 0 && H > 0 && L > 0 && L <= LMAX, "mgnns_sq_mha_layer_bf16_fwd: L=%d unsupported (1..%d)", L, LMAX);
    MG_REQUIRE(H * DK <= QMAX && (H * DK) % 32 == 0, "mgnns_sq_mha_layer_bf16_fwd: n_head=%d unsupported", H);
    MG_REQUIRE(mg_aligned16(bank_bf16) && mg_aligned16(Wp) && mg_aligned16(o_scratch), "mgnns_sq_mha_layer_bf16_fwd: bank/Wp/o must be 16-byte aligned");
    for (int i = 0; i < 6; ++i) MG_REQUIRE(packed[i], "mgnns_sq_mha_layer_bf16_fwd: packed weight %d missing", i);
    MG_REQUIRE(!packed[6] || (qh_next && HK_next > 0), "mgnns_sq_mha_layer_bf16_fwd: next-layer projection incomplete");
    if (B == 0) return 0;
    mg_tail::TailW w;
    w.fc_h = (const unsigned short*)packed[0]; w.fc_l = (const unsigned short*)packed[1];
    w.w1_h = (const unsigned short*)packed[2]; w.w1_l = (const unsigned short*)packed[3];
    w.w2_h = (const unsigned short*)packed[4]; w.w2_l = (const unsigned short*)packed[5];
    w.wq_h = (const unsigned short*)packed[6]; w.wq_l = (const unsigned short*)packed[7];
    w.fc_b = fc_b; w.g1 = ln1_gamma; w.be1 = ln1_beta; w.b1 = b1; w.b2 = b2; w.g2 = ln2_gamma; w.be2 = ln2_beta; w.bq = bq_next;
    const int so = ((H * DK) >> 3) + 2;
    const size_t tail_lds = (size_t)(2 * mg_tail::ROWS * so + 2 * mg_tail::ROWS * mg_tail::SCD) * 16 + 2 * (size_t)mg_tail::ROWS * mg_tail::SD * sizeof(float);
    MG_REQUIRE(tail_lds <= SMEM_BYTES, "mgnns_sq_mha_layer_bf16_fwd: the tail needs %zu B of LDS, the core has %zu", tail_lds, SMEM_BYTES);
    MG_DYN_LDS(sq_mha_layer_bf16_kernel, SMEM_BYTES);
    const int pairs = (H + 1) / 2;
    int gy = 1;
    const int n_cu = mg_cu_count();
    if (n_cu <= 0) return MGNNS_ERR_LAUNCH;
    while (gy < pairs && B * gy < n_cu) gy *= 2;
    if (gy > pairs) gy = pairs;
    const float temp = (float)sqrt((double)dk);
    hipLaunchKernelGGL(sq_mha_layer_bf16_kernel, dim3(B, gy), dim3(NTHR), SMEM_BYTES, (hipStream_t)stream, qh,
                       reinterpret_cast<const unsigned short*>(bank_bf16), mask, B, L, H, reinterpret_cast<const unsigned short*>(Wp), bk,
                       bv, temp, o_scratch, q_in, w, eps, out, HK_next, qh_next, tile_counters);
    MG_CHECK_LAUNCH("mgnns_sq_mha_layer_bf16_fwd");
    return 0;
}
