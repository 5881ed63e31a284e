// Split-bf16 ("bf16x3") form of the fused single-query multi-head attention (submodules.py:55-119, len_q == 1): the reference's
// own formulation -- K and V projected from the memory bank -- at fp32-class accuracy on the bf16 matrix pipe.  Every fp32
// operand travels as hi = bf16(x), lo = bf16(x - hi); a product is THREE v_mfma_f32_16x16x32_bf16 (w_hi x_hi + w_lo x_hi +
// w_hi x_lo, fp32 accumulation: ~2^-16 relative per product); scores, mask, softmax and the probability-weighted sum stay fp32.
// K and V never leave the accumulators.
//
// Geometry (MI355X): one 512-thread workgroup per sample.  The bank's hi and lo images ([L, 320] bf16 each, 640-B rows) do not
// fit LDS together for 196 positions (2 x 208 x 656 B = 273 KB of 160), so the rows are walked in TWO HALVES of at most 112
// (7 tiles of 16): a half's two images are staged by LDS-DMA with a 656-B row stride (41 x 16 B: 16 consecutive rows start in
// 16 different 4-bank groups -- conflict-free ds_read_b128 for the fragment pattern row = lane & 15, chunk = lane >> 4), every
// (head, K or V) x 32-head-dim UNIT runs on that half through the unit queue of sq_mha_bf16.hip (the two waves of a SIMD draw a
// slice's units from an LDS ticket; no barrier inside a half), and the halves are joined by an exact fp32 online-softmax merge:
//   half r, head h:  m_r = max_l s_l,  e_l = exp(s_l - m_r),  z_r = sum_l e_l,  u_r = sum_l e_l V_l        (l in the half)
//   o_h = (u_0 c_0 + u_1 c_1) / (z_0 c_0 + z_1 c_1) + b_v,   c_r = exp(m_r - max(m_0, m_1))   (0 for a half without a live row)
// Masked banks of at most 112 positions (the text bank, T = 100) are one half.  Weight fragments (hi and lo images of the
// fragment-major pack, 1 KiB per fragment) stream from L2 five k-steps ahead; a unit of a half reads 40 KB of them for 420 MFMAs.
// The returned attention (optional) is written as raw scores by the head's softmax wave and normalised in place by the workgroup
// at the end, when both halves' maxima and sums are known.
#include "common.hpp"
#include "tile_bf16.hpp"
#include "sq_mha_util.hpp"
#include "sq_mha_plan.hpp"

#ifdef MG_MHAS_TRACE
// profiling aid (off by default): s_memtime stamps of wave 0 / wave 4 of two workgroups at every phase boundary
__device__ unsigned long long g_mhas_trace[4][64];
#define MGS_STAMP(slot)                                                                             \
    do {                                                                                            \
        if ((threadIdx.x & 255) == 0 && (blockIdx.x == 0 || blockIdx.x == 129) && blockIdx.y == 0 && (slot) < 64)  \
            g_mhas_trace[(blockIdx.x ? 2 : 0) + (threadIdx.x >> 8)][(slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define MGS_STAMP(slot) do { } while (0)
#endif

namespace {

using namespace mg_mha;

constexpr int HT = 7;                   // row tiles of 16 per half
constexpr int LH = HT * 16;             // 112 bank rows per half
constexpr int LMAX = 208;               // two halves cover 224 rows; the row maps (mask bias, attn) are sized for 208 like the other cores
constexpr int KP = 320;                 // model dim padded to 10 k-steps of 32
constexpr int KSTEPS = KP / 32;
constexpr int CH = KP / 8;              // 40 16-byte chunks per row
constexpr int LSTR = 41;                // LDS row stride in chunks (656 B)
constexpr int DK = 128;
constexpr int NTHR = 512;
constexpr int MAXH = 8;                 // heads a workgroup can own
constexpr int FRAG = 1024;              // bytes per weight fragment
constexpr int IMG = LH * LSTR * 16;     // bytes of one staged image (73 472)
static_assert((6 * 16 * LSTR + 9 * 4) * 16 < 65536, "ds_read immediate offsets");

// LDS map
constexpr size_t OFF_LO = IMG;                                            // lo image behind the hi image
constexpr size_t OFF_PART = 2 * (size_t)IMG;                              // float [2][4][LH] partial scores (head parity, slice)
constexpr size_t OFF_P = OFF_PART + 2 * 4 * LH * sizeof(float);           // float [MAXH][LH] e_l = exp(s_l - m_r) of the current half
constexpr size_t OFF_MB = OFF_P + MAXH * LH * sizeof(float);              // float [LMAX] mask bias: 0 or -inf
constexpr size_t OFF_STAT = OFF_MB + LMAX * sizeof(float);                // float [MAXH][2][2]: (m_r, z_r) per local head and half
constexpr size_t OFF_O = OFF_STAT + MAXH * 4 * sizeof(float);             // float [MAXH * DK] u_0 (first half's weighted sums)
constexpr size_t OFF_INT = OFF_O + MAXH * DK * sizeof(float);             // int [16 + 4 MAXH]: live rows, tickets, arrival counts
constexpr size_t OFF_Q = OFF_INT + (16 + 4 * MAXH) * sizeof(int);         // float [MAXH * DK] this sample's projected query
constexpr size_t SMEM_BYTES = OFF_Q + MAXH * DK * sizeof(float);
static_assert(SMEM_BYTES <= 160 * 1024, "LDS");
static_assert(OFF_PART % 16 == 0 && OFF_P % 16 == 0 && OFF_O % 16 == 0 && OFF_Q % 16 == 0, "LDS alignment");

// Wp[img][h][kv][nt][ks][lane][8]: img 0 = hi, 1 = lo of W_kv[h*128 + nt*16 + (lane&15)][ks*32 + (lane>>4)*8 + j]  (0 beyond D)
__global__ __launch_bounds__(256) void pack_kv_weights_split_kernel(const float* __restrict__ Wk, const float* __restrict__ Wv,
                                                                    int H, int D, unsigned short* __restrict__ Wp) {
    const size_t total = (size_t)H * 2 * 8 * KSTEPS * 64;       // fragment-lanes per image
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
        unsigned short vh[8], vl[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = (k0 + j < D) ? W[(size_t)row * D + k0 + j] : 0.f;
            vh[j] = f2bf_t(x);
            vl[j] = f2bf_t(x - bf2f_t(vh[j]));
        }
        uint4 oh, ol;
        oh.x = vh[0] | ((unsigned)vh[1] << 16); oh.y = vh[2] | ((unsigned)vh[3] << 16);
        oh.z = vh[4] | ((unsigned)vh[5] << 16); oh.w = vh[6] | ((unsigned)vh[7] << 16);
        ol.x = vl[0] | ((unsigned)vl[1] << 16); ol.y = vl[2] | ((unsigned)vl[3] << 16);
        ol.z = vl[4] | ((unsigned)vl[5] << 16); ol.w = vl[6] | ((unsigned)vl[7] << 16);
        reinterpret_cast<uint4*>(Wp)[i] = oh;
        reinterpret_cast<uint4*>(Wp)[total + i] = ol;
    }
}

// hi[r, 0:ld] = bf16(x[r, 0:D]), lo[r, 0:ld] = bf16(x - hi), zero padded to ld
__global__ __launch_bounds__(256) void split_pad_bf16_kernel(const float* __restrict__ x, size_t rows, int D, int ld,
                                                             unsigned short* __restrict__ hi, unsigned short* __restrict__ lo) {
    const int c8n = ld / 8;
    const size_t total = rows * (size_t)c8n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / c8n;
        const int c0 = (int)(i - r * c8n) * 8;
        float v[8];
        if (c0 + 8 <= D && (D & 3) == 0) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(x + r * D + c0), b = *reinterpret_cast<const f32x4*>(x + r * D + c0 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (c0 + j < D) ? x[r * D + c0 + j] : 0.f;
        }
        unsigned short vh[8], vl[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            vh[j] = f2bf_t(v[j]);
            vl[j] = f2bf_t(v[j] - bf2f_t(vh[j]));
        }
        uint4 oh, ol;
        oh.x = vh[0] | ((unsigned)vh[1] << 16); oh.y = vh[2] | ((unsigned)vh[3] << 16);
        oh.z = vh[4] | ((unsigned)vh[5] << 16); oh.w = vh[6] | ((unsigned)vh[7] << 16);
        ol.x = vl[0] | ((unsigned)vl[1] << 16); ol.y = vl[2] | ((unsigned)vl[3] << 16);
        ol.z = vl[4] | ((unsigned)vl[5] << 16); ol.w = vl[6] | ((unsigned)vl[7] << 16);
        reinterpret_cast<uint4*>(hi)[i] = oh;
        reinterpret_cast<uint4*>(lo)[i] = ol;
    }
}

// Weight fragments in flight: BD k-steps of (hi, lo) x two column tiles, carried ACROSS unit boundaries -- the last BD k-steps of a
// GEMM request the NEXT unit's first fragments, so an epilogue runs with them on their way (an L2 round trip is 1-2 us, a k-step
// of a 7-tile half 0.35 us).
template <int NMT>
struct Frags {
    static constexpr int BD = 5;                     // must divide KSTEPS (ring slots carry over units)
    static constexpr int TOTAL = KSTEPS * NMT;       // (k-step, row tile) pairs of a GEMM
    static constexpr int RA = TOTAL < 4 ? TOTAL : 4; // bank fragment PAIRS (hi, lo) in flight: requested RA row tiles = 6 RA MFMAs ahead
    uint4 bh[BD][2], bl[BD][2];
};
template <int NMT>
__device__ __forceinline__ void frags_prime_b(Frags<NMT>& f, const WStream& w, int wb, int lo_off) {
#pragma unroll
    for (int d = 0; d < Frags<NMT>::BD; ++d) {
        f.bh[d][0] = wfrag(w, wb + d * FRAG);
        f.bh[d][1] = wfrag(w, wb + (KSTEPS + d) * FRAG);
        f.bl[d][0] = wfrag(w, wb + lo_off + d * FRAG);
        f.bl[d][1] = wfrag(w, wb + lo_off + (KSTEPS + d) * FRAG);
    }
}

// acc[i][j] += W^T[tile j] . X[tile i] over the padded model dim with split operands, for a COMPILE-TIME number of live row tiles
// of the staged half.  Written out like sq_mha_bf16.hip's kv_gemm: bank fragments by inline-asm ds_read_b128 with immediate
// offsets from the two image bases, into a ring that runs RA row tiles ahead; ONE hand-counted s_waitcnt lgkmcnt per row tile
// (six MFMAs); the reads that refill the ring slot follow the MFMAs at once; sched_barrier(0) fences pin the order.
template <int NMT, typename NextStream>
__device__ __forceinline__ void kv_gemm3(f32x4 (&acc)[HT][2], Frags<NMT>& f, unsigned a_h, unsigned a_l, const WStream& w, int wb,
                                         int lo_off, NextStream&& next_stream) {
    constexpr int BD = Frags<NMT>::BD, RA = Frags<NMT>::RA, TOTAL = Frags<NMT>::TOTAL;
    constexpr int HOOK_KS = KSTEPS - BD - 1 > 0 ? KSTEPS - BD - 1 : 0;      // k-step in front of which the next unit is drawn
    u32x4 gh[RA], gl[RA];
    int wb_next = 0;
    auto fetch = [&](auto nc) {
        constexpr int n = decltype(nc)::v;
        constexpr int ks = n / NMT, i = n % NMT;
        gh[n % RA] = mg_lds_read128<(i * 16 * LSTR + ks * 4) * 16>(a_h);
        gl[n % RA] = mg_lds_read128<(i * 16 * LSTR + ks * 4) * 16>(a_l);
    };
    static_for<0, RA>(fetch);
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, KSTEPS>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::v;
        const bf16x8 bh0 = __builtin_bit_cast(bf16x8, f.bh[ks % BD][0]);
        const bf16x8 bh1 = __builtin_bit_cast(bf16x8, f.bh[ks % BD][1]);
        const bf16x8 bl0 = __builtin_bit_cast(bf16x8, f.bl[ks % BD][0]);
        const bf16x8 bl1 = __builtin_bit_cast(bf16x8, f.bl[ks % BD][1]);
        if (ks == HOOK_KS) wb_next = next_stream();
        static_for<0, NMT>([&](auto ic) {
            constexpr int i = decltype(ic)::v;
            constexpr int n0 = ks * NMT + i;
            // pairs requested so far: 0 .. min(n0 + RA, TOTAL) - 1; pair n0 must have landed: two reads per younger pair may stay out
            constexpr int issued = n0 + RA < TOTAL ? n0 + RA : TOTAL;
            mg_lds_wait<2 * (issued - n0 - 1)>();
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 ah = __builtin_bit_cast(bf16x8, gh[n0 % RA]);
            const bf16x8 al = __builtin_bit_cast(bf16x8, gl[n0 % RA]);
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl0, ah, acc[i][0], 0, 0, 0);      // small terms first
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl1, ah, acc[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh0, al, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh1, al, acc[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh0, ah, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh1, ah, acc[i][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (n0 + RA < TOTAL) fetch(IC<n0 + RA>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        if (ks + BD < KSTEPS) {
            f.bh[ks % BD][0] = wfrag(w, wb + (ks + BD) * FRAG);
            f.bh[ks % BD][1] = wfrag(w, wb + (KSTEPS + ks + BD) * FRAG);
            f.bl[ks % BD][0] = wfrag(w, wb + lo_off + (ks + BD) * FRAG);
            f.bl[ks % BD][1] = wfrag(w, wb + lo_off + (KSTEPS + ks + BD) * FRAG);
        } else {                                     // next unit's k-steps 0..BD-1
            f.bh[ks % BD][0] = wfrag(w, wb_next + (ks + BD - KSTEPS) * FRAG);
            f.bh[ks % BD][1] = wfrag(w, wb_next + (ks + BD) * FRAG);
            f.bl[ks % BD][0] = wfrag(w, wb_next + lo_off + (ks + BD - KSTEPS) * FRAG);
            f.bl[ks % BD][1] = wfrag(w, wb_next + lo_off + (ks + BD) * FRAG);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}

struct Ctx {
    unsigned char* smem;
    int B, L, H, lvalid, lo_off;
    const unsigned short* Wp;
    const float* bv;
    float inv_temp;
    float* o;
    float* attn;
};

// One half of the bank rows (rows half * LH ...), NMT live row tiles staged: every unit of this workgroup's heads.
//   K unit   GEMM -> this slice's partial scores of the head -> arrival at the head's count; the LAST of the four slices to arrive
//            runs the head's softmax over the half (one wave, two positions per lane) and publishes e_l and (m_r, z_r)
//   V unit   GEMM -> wait for the head's e_l -> weighted sum u_r -> first half of two: parked in LDS; last half: merged, written
// `first` tells whether weight fragments of this half's first unit still have to be requested (the first half primes them while the
// bank DMA lands).
template <int NMT>
__device__ __forceinline__ void half_body(const Ctx& c, int half, bool two, int& stamp) {
    unsigned char* smem = c.smem;
    float* s_part = reinterpret_cast<float*>(smem + OFF_PART);
    float* s_p = reinterpret_cast<float*>(smem + OFF_P);
    const float* s_mb = reinterpret_cast<const float*>(smem + OFF_MB);
    float* s_stat = reinterpret_cast<float*>(smem + OFF_STAT);
    float* s_o = reinterpret_cast<float*>(smem + OFF_O);
    int* s_int = reinterpret_cast<int*>(smem + OFF_INT);
    int* s_ticket = s_int + 4 + half * 4;               // [4] next unit of a slice, per half
    int* s_kdone = s_int + 16 + half * MAXH;            // [MAXH] slices that delivered their partial scores of local head n
    int* s_smdone = s_int + 16 + 2 * MAXH + half * MAXH;      // [MAXH] 1 = e_l / statistics of local head n published
    const float* s_q = reinterpret_cast<const float*>(smem + OFF_Q);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave & 3;                                        // slice: head dims 32 wq ... 32 wq + 31
    const int b = blockIdx.x, B = c.B, L = c.L, H = c.H;
    const unsigned a_h = mg_lds_addr(smem + ((lane & 15) * LSTR + (lane >> 4)) * 16), a_l = a_h + (unsigned)OFF_LO;
    WStream wsr;
    wsr.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(c.Wp), 0, 0x7fffffff, 0x00027000);
    wsr.voff = lane * 16;
    const int lo_off = c.lo_off;
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
    auto wstream = [&](int t) {                         // byte offset of a unit's first hi fragment for this wave (past the end: a harmless re-read)
        const int h = t < nunits ? head_of(t) : 0;
        return (((h * 2 + ((t >> 1) & 1)) * 8 + wq * 2) * KSTEPS) * FRAG;
    };
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(c.o + (size_t)b * H * DK, 0, H * DK * 4, 0x00027000);
    const __amdgpu_buffer_rsrc_t bv_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.bv), 0, c.bv ? H * DK * 4 : 0, 0x00027000);
    const __amdgpu_buffer_rsrc_t attn_rsrc = __builtin_amdgcn_make_buffer_rsrc(c.attn, 0, c.attn ? 0x7fffffff : 0, 0x00027000);
    // the tiles are computed TRANSPOSED (rows = head dims, columns = bank rows): this lane's accumulator element
    // [i][j][r] is head dim d(j,r) = wq*32 + 16j + 4*(lane>>4) + r of bank row half*LH + 16i + (lane&15)
    const int dbase = wq * 32 + (lane >> 4) * 4;
    const int row0 = half * LH;
    const bool last_half = !two || half == 1;

    int t = draw();
    Frags<NMT> f;
    frags_prime_b<NMT>(f, wsr, wstream(t), lo_off);     // weight fragments on their way while the bank DMA lands
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's LDS-DMA pieces of the half (not tracked by hipcc)
    __syncthreads();                                    // ... and every other wave's
    MGS_STAMP(stamp++);

    while (t < nunits) {
        const int n = (t >> 2) * 2 + (t & 1);           // workgroup-local index of the head
        const int h = head_of(t);
        const bool vunit = (t & 2) != 0;
        int t_next = nunits;
        f32x4 acc[HT][2];
#pragma unroll
        for (int i = 0; i < HT; ++i) {
            acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        kv_gemm3<NMT>(acc, f, a_h, a_l, wsr, wstream(t), lo_off, [&]() { t_next = draw(); return wstream(t_next); });
#pragma unroll
        for (int i = 0; i < NMT; ++i) asm volatile("" : "+v"(acc[i][0]), "+v"(acc[i][1]));      // the MFMAs stay where they were written
        MGS_STAMP(stamp++);

        if (!vunit) {
            // ---- partial scores of this slice's 32 head dims (b_k shifts every score of the head by one constant, which the softmax
            //      cancels: it never enters): in-register over the 8 dims of the lane, then across the four 16-lane groups, four
            //      row tiles per transposing reduction
            const float* qv = s_q + n * DK;
            const f32x4 qd0 = *reinterpret_cast<const f32x4*>(qv + dbase), qd1 = *reinterpret_cast<const f32x4*>(qv + dbase + 16);
            float v[(NMT + 3) / 4 * 4];
#pragma unroll
            for (int i = 0; i < (NMT + 3) / 4 * 4; ++i) {
                float a = 0.f, cc = 0.f;
                if (i < NMT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        a = fmaf(qd0[r], acc[i][0][r], a);
                        cc = fmaf(qd1[r], acc[i][1][r], cc);
                    }
                }
                v[i] = a + cc;
            }
            if (n >= 2) lds_wait_ge(s_smdone + n - 2, 1);           // the slot's previous head has been consumed
            float* part = s_part + ((n & 1) * 4 + wq) * LH;
            const int rt = ((lane >> 4) & 1) * 2 + (lane >> 5);      // row tile (inside a group of 4) of this lane's row
#pragma unroll
            for (int g = 0; g < (NMT + 3) / 4; ++g) {
                const float s4 = rows4_sum4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
                if (4 * g + 3 < NMT || 4 * g + rt < NMT) part[(4 * g + rt) * 16 + (lane & 15)] = s4;
            }
            if (lds_arrive(s_kdone + n, lane) == 3) {
                // ---- last slice of the head: masked scores of the half, its maximum, e_l, their sum; two positions per lane
                const float* sp = s_part + (n & 1) * 4 * LH + lane;
                float sc[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int pos = lane + 64 * j, gpos = row0 + pos;
                    sc[j] = -INFINITY;
                    if (pos < NMT * 16 && gpos < c.lvalid)
                        sc[j] = ((sp[64 * j] + sp[64 * j + LH]) + (sp[64 * j + 2 * LH] + sp[64 * j + 3 * LH])) * c.inv_temp + s_mb[gpos];
                    if (c.attn && pos < NMT * 16 && gpos < L)        // raw scores: normalised in place at the end (finish_attn)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, sc[j]), attn_rsrc, gpos * 4, (h * B + b) * L * 4, 0);
                }
                const float m = wave_max_dpp(fmaxf(sc[0], sc[1]));
                float e[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) e[j] = (sc[j] != -INFINITY) ? __expf(sc[j] - m) : 0.f;
                const float z = wave_sum_dpp(e[0] + e[1]);
                float* prow = s_p + n * LH;
                prow[lane] = e[0];
                if (lane + 64 < LH) prow[lane + 64] = e[1];
                if (lane == 0) {
                    s_stat[(n * 2 + half) * 2] = m;
                    s_stat[(n * 2 + half) * 2 + 1] = z;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(s_smdone + n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
            // ---- u[d] = sum_l e[l] * V[l,d]: e is per column here, 8 dims per lane accumulate in registers over the row tiles, one
            //      16-lane DPP sum per dim at the end.  b_v is added once to the merged output (the probabilities sum to 1).
            lds_wait_ge(s_smdone + n, 1);
            const float* pp = s_p + n * LH + (lane & 15);
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
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                t0[r] = row16_sum(t0[r]);
                t1[r] = row16_sum(t1[r]);
            }
            if ((lane & 15) == 0) {
                float* so = s_o + n * DK + dbase;
                if (!last_half) {                        // first of two halves: park u_0
                    *reinterpret_cast<f32x4*>(so) = t0;
                    *reinterpret_cast<f32x4*>(so + 16) = t1;
                } else {
                    float c1 = 1.0f, zs = s_stat[(n * 2 + half) * 2 + 1];
                    if (two) {
                        const float m0 = s_stat[(n * 2) * 2], z0 = s_stat[(n * 2) * 2 + 1], m1 = s_stat[(n * 2 + 1) * 2];
                        const float m = fmaxf(m0, m1);
                        const float c0 = (m0 != -INFINITY) ? __expf(m0 - m) : 0.f;
                        c1 = (m1 != -INFINITY) ? __expf(m1 - m) : 0.f;
                        zs = z0 * c0 + zs * c1;
                        const f32x4 u0 = *reinterpret_cast<const f32x4*>(so), u1 = *reinterpret_cast<const f32x4*>(so + 16);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            t0[r] = fmaf(u0[r], c0, t0[r] * c1);
                            t1[r] = fmaf(u1[r], c0, t1[r] * c1);
                        }
                    }
                    const float rz = 1.0f / zs;         // all masked: 0 * inf = NaN, like the reference's softmax of -inf
                    const int hoff = h * DK * 4;        // wave-uniform byte offset of the head
                    f32x4 vb0 = {0.f, 0.f, 0.f, 0.f}, vb1 = vb0;
                    if (c.bv) {
                        vb0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bv_rsrc, dbase * 4, hoff, 0));
                        vb1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bv_rsrc, dbase * 4 + 64, hoff, 0));
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        t0[r] = fmaf(t0[r], rz, vb0[r]);
                        t1[r] = fmaf(t1[r], rz, vb1[r]);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, t0), o_rsrc, dbase * 4, hoff, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, t1), o_rsrc, dbase * 4 + 64, hoff, 0);
                }
            }
        }
        MGS_STAMP(stamp++);
        t = t_next;
    }
}

// Stage rows row0 .. row0 + rows - 1 of the sample's hi and lo images by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, every
// piece in flight at once).  A DMA instruction fills 1 KiB of CONTIGUOUS LDS (M0 base + lane*16) from per-lane addresses, so the
// padded [row][41-chunk] image is walked linearly: lanes that fall on the pad chunk of a row are switched off, rows >= L read a
// zero chunk (the zero padding at the end of bank row 0).
__device__ __forceinline__ void stage_half(unsigned char* smem, const uint4* __restrict__ xh, const uint4* __restrict__ xl, int row0,
                                           int rows, int L) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = rows * LSTR;
#pragma unroll
    for (int img = 0; img < 2; ++img) {
        const uint4* xb = img ? xl : xh;
        unsigned char* dst = smem + (img ? OFF_LO : 0);
        for (int pc = wave; pc * 64 < total; pc += NTHR / 64) {
            const int g = pc * 64 + lane;
            const int row = g / LSTR, cidx = g - row * LSTR;
            if (g < total && cidx < CH) {
                const uint4* src = row0 + row < L ? xb + (size_t)(row0 + row) * CH + cidx : xb + (CH - 1);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(uintptr_t)(dst + (size_t)pc * 1024), 16, 0, 0);
            }
        }
    }
}

__device__ __forceinline__ void run_half(const Ctx& c, int n_sel, int half, bool two, int& stamp) {
    switch (n_sel) {
        case 1: half_body<1>(c, half, two, stamp); break;
        case 2: half_body<2>(c, half, two, stamp); break;
        case 4: half_body<4>(c, half, two, stamp); break;
        case 6: half_body<6>(c, half, two, stamp); break;
        default: half_body<HT>(c, half, two, stamp); break;
    }
}
__device__ __forceinline__ int tile_class(int n_mt) { return n_mt <= 1 ? 1 : n_mt <= 2 ? 2 : n_mt <= 4 ? 4 : n_mt <= 6 ? 6 : HT; }

__global__ __launch_bounds__(NTHR) void sq_mha_core_split_kernel(const float* __restrict__ qh, const unsigned short* __restrict__ bank_hi,
                                                                 const unsigned short* __restrict__ bank_lo,
                                                                 const float* __restrict__ mask, int B, int L, int H,
                                                                 const unsigned short* __restrict__ Wp, int lo_off,
                                                                 const float* __restrict__ bv, float temp, float* __restrict__ o,
                                                                 float* __restrict__ attn) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* s_int = reinterpret_cast<int*>(smem + OFF_INT);
    float* s_mb = reinterpret_cast<float*>(smem + OFF_MB);
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const uint4* xh = reinterpret_cast<const uint4*>(bank_hi) + (size_t)b * L * CH;
    const uint4* xl = reinterpret_cast<const uint4*>(bank_lo) + (size_t)b * L * CH;
    int stamp = 1;
    MGS_STAMP(0);

    // ---- live rows, mask bias, the unit queues' counters
    if (tid < 16 + 4 * MAXH) s_int[tid] = (tid == 0 && !mask) ? L : 0;
    __syncthreads();
    {
        int last = 0;
        for (int t = tid; t < LMAX; t += NTHR) {
            const bool live = t < L && (!mask || mask[(size_t)b * L + t] != 0.0f);
            if (mask && live) last = t + 1;
            s_mb[t] = (t < L && !live) ? -INFINITY : 0.0f;
        }
        if (last) atomicMax(s_int, last);
        __syncthreads();
    }
    const int lvalid = s_int[0];
    const bool two = lvalid > LH;
    const int n0 = two ? HT : tile_class((lvalid + 15) >> 4);
    const int n1 = two ? tile_class((lvalid - LH + 15) >> 4) : 0;

    stage_half(smem, xh, xl, 0, n0 * 16, L);
    // this workgroup's heads' query rows -> LDS, by LOCAL head index (visible after the staging barrier in half_body)
    const int pairs = (H + 1) / 2;
    const int npairs = (pairs - (int)blockIdx.y + (int)gridDim.y - 1) / (int)gridDim.y;
    {
        float* s_q = reinterpret_cast<float*>(smem + OFF_Q);
        for (int i = tid * 4; i < npairs * 2 * DK; i += NTHR * 4) {
            const int n = i / DK, h = ((int)blockIdx.y + (n >> 1) * (int)gridDim.y) * 2 + (n & 1);
            f32x4 q = {0.f, 0.f, 0.f, 0.f};
            if (h < H) q = *reinterpret_cast<const f32x4*>(qh + (size_t)b * H * DK + h * DK + (i - n * DK));
            *reinterpret_cast<f32x4*>(s_q + i) = q;
        }
    }
    Ctx c{smem, B, L, H, lvalid, lo_off, Wp, bv, 1.0f / temp, o, attn};
    run_half(c, n0, 0, two, stamp);
    if (two) {
        __syncthreads();                               // every unit of the first half is done with the images
        stage_half(smem, xh, xl, LH, n1 * 16, L);
        run_half(c, n1, 1, true, stamp);
    }
    if (attn) {
        // ---- raw scores -> probabilities, in place: p_l = exp(s_l - m) / z with the merged statistics
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const float* s_stat = reinterpret_cast<const float*>(smem + OFF_STAT);
        for (int i = tid; i < npairs * 2 * L; i += NTHR) {
            const int n = i / L, pos = i - n * L;
            const int h = ((int)blockIdx.y + (n >> 1) * (int)gridDim.y) * 2 + (n & 1);
            if (h >= H) continue;
            float* a = attn + ((size_t)h * B + b) * L + pos;
            float m = s_stat[(n * 2) * 2], z = s_stat[(n * 2) * 2 + 1];
            if (two) {
                const float m1 = s_stat[(n * 2 + 1) * 2], z1 = s_stat[(n * 2 + 1) * 2 + 1];
                const float mm = fmaxf(m, m1);
                z = z * ((m != -INFINITY) ? __expf(m - mm) : 0.f) + z1 * ((m1 != -INFINITY) ? __expf(m1 - mm) : 0.f);
                m = mm;
            }
            float p = 0.f;
            if (pos < lvalid) {
                const float s = *a;
                p = (s != -INFINITY) ? __expf(s - m) * (1.0f / z) : 0.f * (1.0f / z);
            } else {
                p = 0.f * (1.0f / z);                   // NaN when every position is masked (z == 0), like the reference
            }
            *a = p;
        }
    }
}


// =====================================================================================================================
// Grouped form for MASKED banks of at most 112 positions (the text bank: mean 16 live rows of 100, MODEL:509-527).  A workgroup of
// the per-sample kernel streams its heads' hi + lo weights (2.6 MB for eight heads) whatever its sample's rows: 256 samples = 666 MB
// of L2 traffic for ~300 tiles of live rows (51-63 us per launch).  Here the samples of a GROUP -- whole 16-row tiles each, at most 7
// tiles and 7 samples (plan: sq_mha_plan.hpp with ALIGN 16 / ROWS 112 / SAMP 7) -- share one staging of the two images and one pass
// of the weight stream; a workgroup owns one head pair of one group (persistent over the groups).  K units: the score of a row uses
// ITS sample's query (q rows of the group's samples in LDS).  Softmax: a 16-lane DPP row is a tile, tile maxima / sums are joined
// per sample through eight LDS words.  V units: the weighted sum is flushed (row sum, 1 / z, + b_v, store) behind the last tile of
// every sample -- compile-time tile index, wave-uniform flush mask.  No returned attention (the launcher takes the per-sample kernel
// when it is asked for).
constexpr int GS = 7;                                                     // samples a group holds
constexpr int GHL = 2;                                                    // heads a workgroup owns (one pair)
constexpr size_t G_OFF_PART = 2 * (size_t)IMG;                            // float [2][4][LH] partial scores (head of the pair, slice)
constexpr size_t G_OFF_P = G_OFF_PART + 2 * 4 * LH * sizeof(float);       // float [GHL][LH] e_l
constexpr size_t G_OFF_MB = G_OFF_P + GHL * LH * sizeof(float);           // float [LH] mask bias of the staged rows
constexpr size_t G_OFF_Q = G_OFF_MB + LH * sizeof(float);                 // float [GS][GHL][DK] projected queries of the group's samples
constexpr size_t G_OFF_Z = G_OFF_Q + GS * GHL * DK * sizeof(float);       // float [GS][GHL] softmax sums (+ pad)
constexpr size_t G_OFF_TM = G_OFF_Z + 16 * sizeof(float);                 // float [GHL][8 tile maxima | 8 tile sums]: the two heads' softmax waves run side by side
constexpr size_t G_OFF_INT = G_OFF_TM + GHL * 16 * sizeof(float);               // int [48]: see grouped_kernel
constexpr size_t G_SMEM_BYTES = G_OFF_INT + 48 * sizeof(int);
static_assert(G_SMEM_BYTES <= 160 * 1024 && G_OFF_Q % 16 == 0 && G_OFF_P % 16 == 0, "LDS (grouped)");

struct GCtx {
    unsigned char* smem;
    int B, L, H, lo_off, b0, h0, nheads, flush;     // first sample of the group, first head of the pair, heads of the pair, flush mask
    const unsigned short* Wp;
    const float* bv;
    float inv_temp;
    float* o;
};

template <int NMT>
__device__ __forceinline__ void group_body(const GCtx& c) {
    unsigned char* smem = c.smem;
    float* s_part = reinterpret_cast<float*>(smem + G_OFF_PART);
    float* s_p = reinterpret_cast<float*>(smem + G_OFF_P);
    const float* s_mb = reinterpret_cast<const float*>(smem + G_OFF_MB);
    const float* s_q = reinterpret_cast<const float*>(smem + G_OFF_Q);
    float* s_z = reinterpret_cast<float*>(smem + G_OFF_Z);
    float* s_tm_all = reinterpret_cast<float*>(smem + G_OFF_TM);
    int* s_int = reinterpret_cast<int*>(smem + G_OFF_INT);
    int* s_ticket = s_int;                              // [4] next unit of a slice
    int* s_kdone = s_int + 4;                           // [2] slices that delivered their partial scores of head n
    int* s_smdone = s_int + 6;                          // [2] 1 = e_l / sums of head n published
    const int* s_tsamp = s_int + 16;                    // [8] group-local sample of a tile
    const int* s_tfirst = s_int + 24;                   // [8] first tile of that sample
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave & 3;
    const int H = c.H;
    const unsigned a_h = mg_lds_addr(smem + ((lane & 15) * LSTR + (lane >> 4)) * 16), a_l = a_h + (unsigned)OFF_LO;
    WStream wsr;
    wsr.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(c.Wp), 0, 0x7fffffff, 0x00027000);
    wsr.voff = lane * 16;
    const int lo_off = c.lo_off;
    // tickets K(h0) K(h1) V(h0) V(h1); a head beyond the pair's count is skipped.  (Opaque to the compiler: with a literal 4 it
    //  unrolls the unit loop into four GEMM instances -- 200 spilled registers.)
    int nunits = 4;
    asm volatile("" : "+s"(nunits));
    auto draw = [&]() {
        int t;
        do {
            int v = 0;
            if (lane == 0) v = __hip_atomic_fetch_add(s_ticket + wq, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            t = __builtin_amdgcn_readfirstlane(v);
        } while (t < nunits && (t & 1) >= c.nheads);
        return t;
    };
    auto wstream = [&](int t) {
        const int h = c.h0 + (t < nunits ? (t & 1) : 0);
        return (((h * 2 + ((t >> 1) & 1)) * 8 + wq * 2) * KSTEPS) * FRAG;
    };
    const __amdgpu_buffer_rsrc_t bv_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.bv), 0, c.bv ? H * DK * 4 : 0, 0x00027000);
    const int dbase = wq * 32 + (lane >> 4) * 4;
    // group-local sample of every tile: wave-uniform values (scalar registers)
    int ts[HT];
#pragma unroll
    for (int i = 0; i < HT; ++i) ts[i] = __builtin_amdgcn_readfirstlane(s_tsamp[i < NMT ? i : 0]);

    int t = draw();
    Frags<NMT> f;
    frags_prime_b<NMT>(f, wsr, wstream(t), lo_off);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's LDS-DMA pieces (not tracked by hipcc)
    __syncthreads();

    while (t < nunits) {
        const int n = t & 1;                            // head of the pair
        const int h = c.h0 + n;
        const bool vunit = (t & 2) != 0;
        int t_next = nunits;
        f32x4 acc[HT][2];
#pragma unroll
        for (int i = 0; i < HT; ++i) {
            acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        kv_gemm3<NMT>(acc, f, a_h, a_l, wsr, wstream(t), lo_off, [&]() { t_next = draw(); return wstream(t_next); });
#pragma unroll
        for (int i = 0; i < NMT; ++i) asm volatile("" : "+v"(acc[i][0]), "+v"(acc[i][1]));

        if (!vunit) {
            // ---- partial scores: tile i against the query of ITS sample
            float v[(NMT + 3) / 4 * 4];
#pragma unroll
            for (int i = 0; i < (NMT + 3) / 4 * 4; ++i) {
                float a = 0.f, cc = 0.f;
                if (i < NMT) {
                    // (a tile of the class behind the group's rows has no sample, ts = -1: it reads sample 0's query -- its rows
                    //  are zero chunks and its mask bias -inf, so any FINITE query gives e = 0; -1 would index in front of s_q)
                    const float* qv = s_q + ((ts[i] < 0 ? 0 : ts[i]) * GHL + n) * DK + dbase;
                    const f32x4 qd0 = *reinterpret_cast<const f32x4*>(qv), qd1 = *reinterpret_cast<const f32x4*>(qv + 16);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        a = fmaf(qd0[r], acc[i][0][r], a);
                        cc = fmaf(qd1[r], acc[i][1][r], cc);
                    }
                }
                v[i] = a + cc;
            }
            float* part = s_part + (n * 4 + wq) * LH;
            const int rt = ((lane >> 4) & 1) * 2 + (lane >> 5);
#pragma unroll
            for (int g = 0; g < (NMT + 3) / 4; ++g) {
                const float s4 = rows4_sum4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
                if (4 * g + 3 < NMT || 4 * g + rt < NMT) part[(4 * g + rt) * 16 + (lane & 15)] = s4;
            }
            if (lds_arrive(s_kdone + n, lane) == 3) {
                // ---- last slice of the head: a 16-lane row of the wave is a tile (two tiles per lane group: rows lane, lane + 64);
                //      tile maxima / sums through eight LDS words each, joined per sample
                const float* sp = s_part + n * 4 * LH + lane;
                float* s_tm = s_tm_all + n * 16;
                float sc[2];
                int smp[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int pos = lane + 64 * j, T = pos >> 4;
                    smp[j] = T < NMT ? s_tsamp[T] : -1;
                    sc[j] = -INFINITY;
                    if (pos < NMT * 16)
                        sc[j] = ((sp[64 * j] + sp[64 * j + LH]) + (sp[64 * j + 2 * LH] + sp[64 * j + 3 * LH])) * c.inv_temp + s_mb[pos];
                    const float tm = row16_max(sc[j]);
                    if ((lane & 15) == 0 && T < 8) s_tm[T] = tm;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                float e[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float m = -INFINITY;
#pragma unroll
                    for (int k = 0; k < NMT; ++k) m = (ts[k] == smp[j]) ? fmaxf(m, s_tm[k]) : m;
                    e[j] = (sc[j] != -INFINITY) ? __expf(sc[j] - m) : 0.f;
                    const float tsum = row16_sum(e[j]);
                    const int T = (lane + 64 * j) >> 4;
                    if ((lane & 15) == 0 && T < 8) s_tm[8 + T] = tsum;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                float* prow = s_p + n * LH;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int pos = lane + 64 * j, T = pos >> 4;
                    if (pos < LH) prow[pos] = e[j];
                    if (T < NMT && (lane & 15) == 0 && s_tfirst[T] == T) {      // first tile of its sample: the sample's sum
                        float z = 0.f;
#pragma unroll
                        for (int k = 0; k < NMT; ++k) z += (ts[k] == smp[j]) ? s_tm[8 + k] : 0.f;
                        s_z[smp[j] * GHL + n] = z;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(s_smdone + n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
            // ---- u[d] = sum_l e[l] V[l, d] per SAMPLE: flushed behind the last tile of every sample
            f32x4 vb0 = {0.f, 0.f, 0.f, 0.f}, vb1 = vb0;
            const int hoff = h * DK * 4;
            if (c.bv) {
                vb0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bv_rsrc, dbase * 4, hoff, 0));
                vb1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bv_rsrc, dbase * 4 + 64, hoff, 0));
            }
            lds_wait_ge(s_smdone + n, 1);
            const float* pp = s_p + n * LH + (lane & 15);
            f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
            static_for<0, NMT>([&](auto ic) {
                constexpr int i = decltype(ic)::v;
                const float p = pp[i * 16];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    t0[r] = fmaf(p, acc[i][0][r], t0[r]);
                    t1[r] = fmaf(p, acc[i][1][r], t1[r]);
                }
                if ((c.flush >> i) & 1) {                  // (wave-uniform)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        t0[r] = row16_sum(t0[r]);
                        t1[r] = row16_sum(t1[r]);
                    }
                    if ((lane & 15) == 0) {
                        const float rz = 1.0f / s_z[ts[i] * GHL + n];      // all masked: 0 * inf = NaN, like the reference
                        f32x4 o0, o1;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            o0[r] = fmaf(t0[r], rz, vb0[r]);
                            o1[r] = fmaf(t1[r], rz, vb1[r]);
                        }
                        float* op = c.o + ((size_t)(c.b0 + ts[i]) * H + h) * DK + dbase;
                        *reinterpret_cast<f32x4*>(op) = o0;
                        *reinterpret_cast<f32x4*>(op + 16) = o1;
                    }
                    t0 = f32x4{0.f, 0.f, 0.f, 0.f};
                    t1 = t0;
                }
            });
        }
        t = t_next;
    }
}

__global__ __launch_bounds__(NTHR) void sq_mha_core_split_grouped_kernel(const float* __restrict__ qh, const unsigned short* __restrict__ bank_hi,
                                                                         const unsigned short* __restrict__ bank_lo,
                                                                         const float* __restrict__ mask, int B, int L, int H,
                                                                         const unsigned short* __restrict__ Wp, int lo_off,
                                                                         const float* __restrict__ bv, float temp, float* __restrict__ o,
                                                                         const int* __restrict__ plan, int* __restrict__ status) {
    // a plan of the other kind (the packed bf16 kernel's: same size, up to 16 samples / 128 rows per group) or of another batch would
    // index past the 48-int maps below: nothing is trusted, nothing is written, the library's status word says so
    if (plan[2] != mg_plan::kind_word(16, LH, GS) || plan[1] != B) {
        if (threadIdx.x == 0 && status) __hip_atomic_store(status, MGNNS_STATUS_BAD_PLAN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* s_int = reinterpret_cast<int*>(smem + G_OFF_INT);
    float* s_mb = reinterpret_cast<float*>(smem + G_OFF_MB);
    float* s_q = reinterpret_cast<float*>(smem + G_OFF_Q);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ng = plan[0];
    const int h0 = (int)blockIdx.y * GHL, nheads = H - h0 < GHL ? H - h0 : GHL;
    for (int g = blockIdx.x; g < ng; g += gridDim.x) {
        const int b0 = plan[mg_plan::PLAN_HDR + 4 * g];
        int ns = plan[mg_plan::PLAN_HDR + 4 * g + 1], rows = plan[mg_plan::PLAN_HDR + 4 * g + 2];
        ns = ns > GS ? GS : ns;                          // (what the maps below hold, whatever the plan says)
        rows = rows > LH ? LH : rows;
        const int nt = rows >> 4;
        // ---- the group's tile maps, counters (s_int: [0..3] tickets, [4..5] kdone, [6..7] smdone, [8] flush mask, [16..23] sample of
        //      a tile, [24..31] first tile of that sample, [32..38] first row of a sample, [40..46] its live rows)
        if (tid < 48) s_int[tid] = (tid >= 16 && tid < 24) ? -1 : 0;      // (a tile of the class behind the group's rows belongs to nobody)
        __syncthreads();
        if (tid < ns) {
            const int off = plan[mg_plan::PLAN_HDR + 4 * B + 2 * (b0 + tid)], lv = plan[mg_plan::PLAN_HDR + 4 * B + 2 * (b0 + tid) + 1];
            const int ntl = lv <= 16 ? 1 : (lv + 15) >> 4;
            s_int[32 + tid] = off;
            s_int[40 + tid] = lv;
            for (int k = 0; k < ntl; ++k) {
                s_int[16 + (off >> 4) + k] = tid;
                s_int[24 + (off >> 4) + k] = off >> 4;
            }
            atomicOr(&s_int[8], 1 << ((off >> 4) + ntl - 1));             // the sample's last tile: flush
        }
        __syncthreads();
        const int flush = s_int[8];
        // ---- stage the group's rows (LDS row R = tile R >> 4 of sample s_tsamp: its row R - 16 first) + mask bias + queries
        {
            // every row of the tile-count CLASS the GEMM runs: rows behind the group's (and behind a sample's L) read a zero chunk
            // (the padding at the end of a bank row) -- stale LDS there would put NaN bits into scores that -inf cannot mask
            const int total = tile_class(nt) * 16 * LSTR;
#pragma unroll
            for (int img = 0; img < 2; ++img) {
                const uint4* xg = reinterpret_cast<const uint4*>(img ? bank_lo : bank_hi);
                unsigned char* dst = smem + (img ? OFF_LO : 0);
                for (int pc = wave; pc * 64 < total; pc += NTHR / 64) {
                    const int gi = pc * 64 + lane;
                    const int R = gi / LSTR, cidx = gi - R * LSTR;
                    if (gi < total && cidx < CH) {
                        const int T = R >> 4, sm = R < rows ? s_int[16 + T] : 0, r = R < rows ? R - 16 * s_int[24 + T] : L;
                        const uint4* xb = xg + (size_t)(b0 + sm) * L * CH;
                        const uint4* src = r < L ? xb + (size_t)r * CH + cidx : xb + (CH - 1);
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                         (__attribute__((address_space(3))) void*)(uintptr_t)(dst + (size_t)pc * 1024), 16, 0, 0);
                    }
                }
            }
            for (int R = tid; R < LH; R += NTHR) {
                float mbv = -INFINITY;
                if (R < rows) {
                    const int T = R >> 4, sm = s_int[16 + T], r = R - 16 * s_int[24 + T];
                    if (r < L && mask[(size_t)(b0 + sm) * L + r] != 0.0f) mbv = 0.0f;
                }
                s_mb[R] = mbv;
            }
            for (int i = tid * 4; i < ns * GHL * DK; i += NTHR * 4) {
                const int sm = i / (GHL * DK), rem = i - sm * (GHL * DK), n = rem / DK, d = rem - n * DK;
                f32x4 q = {0.f, 0.f, 0.f, 0.f};
                if (n < nheads) q = *reinterpret_cast<const f32x4*>(qh + ((size_t)(b0 + sm) * H + h0 + n) * DK + d);
                *reinterpret_cast<f32x4*>(s_q + i) = q;
            }
        }
        GCtx c{smem, B, L, H, lo_off, b0, h0, nheads, flush, Wp, bv, 1.0f / temp, o};
        switch (tile_class(nt)) {
            case 1: group_body<1>(c); break;
            case 2: group_body<2>(c); break;
            case 4: group_body<4>(c); break;
            case 6: group_body<6>(c); break;
            default: group_body<HT>(c); break;
        }
        __syncthreads();                               // every unit of this group is done with LDS before the next one stages
    }
}

__global__ __launch_bounds__(1024) void sq_mha_split_plan_kernel(const float* __restrict__ mask, int B, int L, int* __restrict__ plan) {
    extern __shared__ int s_plan[];
    mg_plan::build<1024, 16, LH, GS>(mask, B, L, plan, s_plan);
}

}  // namespace

#ifdef MG_MHAS_TRACE
extern "C" int mgnns_debug_mhas_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mhas_trace), sizeof(unsigned long long) * 4 * 64) == hipSuccess ? 0 : 1;
}
#endif

extern "C" size_t mgnns_sq_mha_split_packed_weight_bytes(int H) { return (size_t)2 * H * 2 * 8 * KSTEPS * 64 * 16; }

extern "C" int mgnns_sq_mha_pack_weights_split(const float* Wk, const float* Wv, int H, int dk, int D, void* Wp,
                                               mgnns_stream_t stream) {
    MG_REQUIRE(Wk && Wv && Wp, "mgnns_sq_mha_pack_weights_split: null pointer");
    MG_REQUIRE(dk == DK && H > 0 && D > 0 && D <= KP, "mgnns_sq_mha_pack_weights_split: unsupported dk=%d D=%d", dk, D);
    MG_REQUIRE(mg_aligned16(Wp), "mgnns_sq_mha_pack_weights_split: Wp must be 16-byte aligned");
    const size_t total = (size_t)H * 2 * 8 * KSTEPS * 64;
    hipLaunchKernelGGL(pack_kv_weights_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Wk,
                       Wv, H, D, reinterpret_cast<unsigned short*>(Wp));
    MG_CHECK_LAUNCH("mgnns_sq_mha_pack_weights_split");
    return 0;
}

extern "C" int mgnns_split_pad_bf16(const float* x, int64_t rows, int D, int ld, void* hi, void* lo, mgnns_stream_t stream) {
    MG_REQUIRE(x && hi && lo, "mgnns_split_pad_bf16: null pointer");
    MG_REQUIRE(rows >= 0 && D > 0 && ld >= D && ld % 8 == 0, "mgnns_split_pad_bf16: bad dims D=%d ld=%d", D, ld);
    MG_REQUIRE(mg_aligned16(x) && mg_aligned16(hi) && mg_aligned16(lo), "mgnns_split_pad_bf16: x / hi / lo must be 16-byte aligned");
    if (rows == 0) return 0;
    const size_t total = (size_t)rows * (ld / 8);
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(split_pad_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (size_t)rows, D, ld,
                       reinterpret_cast<unsigned short*>(hi), reinterpret_cast<unsigned short*>(lo));
    MG_CHECK_LAUNCH("mgnns_split_pad_bf16");
    return 0;
}

extern "C" int mgnns_sq_mha_split_plan(const float* mask, int B, int L, int32_t* plan, mgnns_stream_t stream) {
    MG_REQUIRE(mask && plan, "mgnns_sq_mha_split_plan: null pointer");
    MG_REQUIRE(B >= 0 && B <= mg_plan::MAX_B && L > 0 && L <= LH, "mgnns_sq_mha_split_plan: B=%d (<= %d), L=%d (1..%d) unsupported", B,
               mg_plan::MAX_B, L, LH);
    const size_t lds = mg_plan::lds_bytes(B);
    MG_DYN_LDS(sq_mha_split_plan_kernel, lds);
    hipLaunchKernelGGL(sq_mha_split_plan_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, mask, B, L, plan);
    MG_CHECK_LAUNCH("mgnns_sq_mha_split_plan");
    return 0;
}

extern "C" int mgnns_sq_mha_core_split_fwd(const float* qh, const void* bank_hi, const void* bank_lo, const float* mask, int B,
                                           int L, int ld, int H, int dk, const void* Wp, const float* bk, const float* bv,
                                           float* o, float* attn, const int32_t* plan, mgnns_stream_t stream) {
    (void)bk;                                           // softmax-invariant (see half_body)
    MG_REQUIRE(qh && bank_hi && bank_lo && Wp && o, "mgnns_sq_mha_core_split_fwd: null pointer");
    MG_REQUIRE(dk == DK, "mgnns_sq_mha_core_split_fwd: d_kv=%d unsupported (128 only)", dk);
    MG_REQUIRE(ld == KP, "mgnns_sq_mha_core_split_fwd: bank row length %d must be %d (bf16 hi / lo images, zero padded)", ld, KP);
    MG_REQUIRE(B >= 0 && H > 0 && L > 0 && L <= LMAX, "mgnns_sq_mha_core_split_fwd: L=%d unsupported (1..%d)", L, LMAX);
    MG_REQUIRE((double)H * B * L * 4 < 2147483648.0, "mgnns_sq_mha_core_split_fwd: attn output beyond 2 GiB (B=%d)", B);
    MG_REQUIRE(H <= 2 * MAXH, "mgnns_sq_mha_core_split_fwd: n_head=%d unsupported (<= %d)", H, 2 * MAXH);
    MG_REQUIRE(mg_aligned16(bank_hi) && mg_aligned16(bank_lo) && mg_aligned16(Wp) && mg_aligned16(qh) && mg_aligned16(o),
               "mgnns_sq_mha_core_split_fwd: qh / banks / Wp / o must be 16-byte aligned");
    MG_REQUIRE(!plan || (mask && L <= LH && !attn), "mgnns_sq_mha_core_split_fwd: a group plan needs a mask, L <= %d and no attn output", LH);
    if (B == 0) return 0;
    if (plan) {
        // grouped masked form: workgroup = (group of samples, head pair), persistent over the groups: a third of the batch is more
        // groups than MVSA-like lengths make (59 for 256 samples); longer batches of long documents loop
        MG_DYN_LDS(sq_mha_core_split_grouped_kernel, G_SMEM_BYTES);
        int gx = (B + 2) / 3;
        if (gx < 1) gx = 1;
        const float temp_g = (float)sqrt((double)dk);
        const int lo_g = (int)(mgnns_sq_mha_split_packed_weight_bytes(H) / 2);
        hipLaunchKernelGGL(sq_mha_core_split_grouped_kernel, dim3(gx, (H + GHL - 1) / GHL), dim3(NTHR), G_SMEM_BYTES, (hipStream_t)stream, qh,
                           reinterpret_cast<const unsigned short*>(bank_hi), reinterpret_cast<const unsigned short*>(bank_lo), mask, B, L,
                           H, reinterpret_cast<const unsigned short*>(Wp), lo_g, bv, temp_g, o, plan, mg_status_word());
        MG_CHECK_LAUNCH("mgnns_sq_mha_core_split_fwd(grouped)");
        return 0;
    }
    MG_DYN_LDS(sq_mha_core_split_kernel, SMEM_BYTES);
    // one workgroup per sample owns all head pairs (at most MAXH heads) when the batch fills the chip; small batches split the pairs
    const int pairs = (H + 1) / 2;
    int gy = 1;
    while (gy < pairs && (pairs + gy - 1) / gy * 2 > MAXH) gy *= 2;
    const int n_cu = mg_cu_count();
    if (n_cu <= 0) return MGNNS_ERR_LAUNCH;
    while (gy < pairs && B * gy < n_cu) gy *= 2;
    // a MASKED bank (the text bank: mean 16 live rows of 100) has no packed form here: the one full-length document of a batch holds
    // the launch open with all its 16 H units on one workgroup -- two workgroups per sample even when the batch fills the chip
    // (B = 256, L = 100, H = 8: 62.7 / 51.3 / 54.4 us with 1 / 2 / 4; every workgroup streams its heads' weights whatever its rows, so
    // more than two only multiplies workgroups); MGNNS_MHA_SPLIT_MASKED overrides
    if (mask) {
        if (gy < 2 && pairs >= 2) gy = 2;
        if (const int e = mg_env_int("MGNNS_MHA_SPLIT_MASKED", 0, 5)) gy = e;
    }
    if (const int e = mg_env_int("MGNNS_MHA_SPLIT", 0, 4)) gy = e > gy ? e : gy;        // measurement knob: workgroups per sample
    while (gy < pairs && (pairs + gy - 1) / gy * 2 > MAXH) gy *= 2;
    if (gy > pairs) gy = pairs;
    const float temp = (float)sqrt((double)dk);
    const int lo_off = (int)(mgnns_sq_mha_split_packed_weight_bytes(H) / 2);
    hipLaunchKernelGGL(sq_mha_core_split_kernel, dim3(B, gy), dim3(NTHR), SMEM_BYTES, (hipStream_t)stream, qh,
                       reinterpret_cast<const unsigned short*>(bank_hi), reinterpret_cast<const unsigned short*>(bank_lo), mask, B, L,
                       H, reinterpret_cast<const unsigned short*>(Wp), lo_off, bv, temp, o, attn);
    MG_CHECK_LAUNCH("mgnns_sq_mha_core_split_fwd");
    return 0;
}
