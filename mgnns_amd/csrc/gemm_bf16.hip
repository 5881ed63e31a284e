// Dense bf16 GEMM on v_mfma_f32_16x16x32_bf16, fp32 accumulation and output (BASELINE configs[4] (i): the dense
// [N,N] adjacency path of the label / vocabulary graph GCN, UTIL:421-426 `adj @ support`, and any large X.W):
//   C[M,N] = act(A[M,K] . Bt[N,K]^T + bias),  A and Bt bf16, K-contiguous rows of Kp elements (Kp % 64 == 0, zero padded)
//
// Workgroup tile 256 x 128, eight compute waves 4 x 2 = two per SIMD (wave tile 64 x 64 = 4 x 4 MFMA tiles; with one wave per
// SIMD and a 128 x 64 wave tile the MFMAs issued at ~36 cycles instead of ~17 next to the fragment reads) plus four producer
// waves that issue the LDS-DMA stream (persistent workgroups, see the kernel),
// BK = 64.  Both operand tiles go global -> LDS by LDS-DMA (no VGPR round trip) through a three-stage ring (48 KB
// per stage: rows of 8 16-B chunks, chunk index XOR ((row >> 1) & 7) so that the ds_read_b128 fragment pattern -- 16 rows
// x one chunk column -- is bank-conflict free without padding, which LDS-DMA could not write); one barrier per
// BK slice.  Fragments are double buffered in registers across the two k-steps of a slice.  Tiles are computed
// transposed (A operand = Bt fragment) so a lane ends up with four consecutive C columns = one 16-byte store.
//
// Workgroup -> tile map is XCD aware (blockIdx & 7 = XCD): the 32 workgroups that run together on an XCD form a
// super tile of (32 / column tiles) row blocks x all column tiles and march along K together, so an A byte is
// fetched from HBM once per XCD and a Bt byte once per super row (for M = K = 10 000, N = 1024: ~0.4 GB instead of
// 1.6 GB + 1.6 GB with a row-major tile order).
#include "common.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int TM = 256, TN = 128, BK = 64, NSTAGE = 3;
constexpr int A_BYTES = TM * BK * 2, B_BYTES = TN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;   // 32 KB + 16 KB = 48 DMA pieces
constexpr size_t SMEM_BYTES = (size_t)NSTAGE * STAGE_BYTES;

// y[c, 0:ld] = bf16(x[0:rows, c]) (transpose + cast, zero padded to ld): the K-contiguous Bt operand from a [K,N] matrix
__global__ __launch_bounds__(256) void transpose_cast_bf16_kernel(const float* __restrict__ x, int rows, int cols, int ld,
                                                                  unsigned short* __restrict__ y) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? x[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;                      // out row = c, out col = r
        if (c < cols && r < ld) {
            unsigned int u = __float_as_uint(tile[tx][i]);
            u += 0x7FFFu + ((u >> 16) & 1u);
            y[(size_t)c * ld + r] = (unsigned short)(u >> 16);
        }
    }
}

// Persistent, warp specialised (the structure measured on the convolutions, conv_bf16.hip): one workgroup per CU walks its
// tiles in XCD-aware order; waves 8..11 (one per SIMD) only issue the LDS-DMA pieces of the slice stream -- the slices of ALL
// the workgroup's tiles, one after the other, through the three-stage ring -- and wait for them; waves 0..7 only read
// fragments, issue MFMAs and store.  (With the DMA issue inside the compute waves each of them stalled ~1.2 k cycles per slice
// in the backed-up vector-memory path against ~0.7 k cycles of MFMA work.)
constexpr int NTHR_WS = 768, NPROD = 4, A_PIECES = A_BYTES / 1024, B_PIECES = B_BYTES / 1024;
constexpr int APP = A_PIECES / NPROD, BPP = B_PIECES / NPROD, PPP = APP + BPP;     // 8 + 4 pieces per producer and slice

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

// (T, full, rem, f) of an XCD: shared by the GEMM kernel, the fix-up kernel and the host launcher
__host__ __device__ inline void mg_gemm_split(int xcd, int nrb, int nct, int rps, int jmax, int W, int nk, int& T, int& full, int& rem, int& f) {
    if (rps == 0) {
        const int cpx = jmax / nrb;
        int cts = nct - xcd * cpx;
        cts = cts < 0 ? 0 : (cts > cpx ? cpx : cts);
        T = cts * nrb;
    } else {
        T = ((xcd + 1) * nrb / 8 - xcd * nrb / 8) * nct;
    }
    full = T / W;
    rem = T - full * W;
    f = 0;
    // only a last round that is at most a QUARTER full is split: the tile stream of a full chip is bound by the L2 -> LDS DMA
    // traffic (a round of 256 CUs takes ~200 us, the 64-tile remainder of 10 000 x 1024 alone on 64 CUs ~95), so halves gain
    // nothing (measured: 10 000 x 2048, f = 2: 473 us split, 451 whole; 10 000 x 1024, f = 4: 275 split, 295 whole)
    if (rem > 0 && 4 * rem <= W) {
        f = W / rem;
        if (f > 8) f = 8;
        if (f * 8 > nk) f = 0;                          // a part needs >= 8 slices: with K = 320 (5 slices per tile) the fix-up launch
    }                                                   // and its 33 MB of partial sums cost more than the round they save
}

__global__ __launch_bounds__(NTHR_WS) void gemm_bf16_nt_kernel(const unsigned short* __restrict__ A,
                                                               const unsigned short* __restrict__ Bt, int M, int N, int Kp,
                                                               const float* __restrict__ bias, float* __restrict__ C, int ldc,
                                                               int act, int nrb, int nct, int rps, int jmax,
                                                               const int32_t* __restrict__ m_dev, int c_bf16,
                                                               float* __restrict__ ws_part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nrb_host = nrb;                                      // (the column-split map is laid out for the host's row-block count)
    if (m_dev) {                                                   // row count produced on the device (ragged batches): M is its bound
        const int md = *m_dev;
        M = md < M ? (md < 0 ? 0 : md) : M;
        nrb = (M + TM - 1) / TM;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, jj0 = blockIdx.x >> 3, jstep = gridDim.x >> 3;
    const int nk = Kp / BK;
    // virtual tile j of this XCD -> (m0, n0); false beyond its last tile.
    // rps != 0: XCD x owns the row blocks [x nrb / 8, (x + 1) nrb / 8) -- a balanced split (rounds 1-3 dealt out whole "super rows"
    //   of 32 / nct row blocks: 40 row blocks = 10 super rows = TWO for XCDs 0 and 1, one for the others, i.e. the chip waited for
    //   two XCDs to run a second round) -- tile j = (row block j / nct of the range, column tile j % nct): the workgroups of a
    //   round share a few row blocks of A and all of Bt's column tiles in the XCD's L2.
    // rps == 0: fewer row blocks than XCDs -- e.g. a 512-row read-out against 10 000 columns -- so the XCDs split the COLUMN
    //   tiles instead: XCD x takes column tiles x * cpx .. + cpx - 1 with every row block, the row blocks of a column tile next to
    //   each other (with a row-block map only two of the eight XCDs had work: 88 us instead of 35 for that read-out)
    const int rb0 = xcd * nrb / 8, rbn = (xcd + 1) * nrb / 8 - rb0;
    auto decode = [&](int j, int& m0, int& n0) {
        if (rps == 0) {
            const int cpx = jmax / nrb_host;
            const int ct = xcd * cpx + j / nrb_host, rb = j % nrb_host;
            m0 = rb * TM;
            n0 = ct * TN;
            return j < jmax && ct < nct && rb < nrb;
        }
        m0 = (rb0 + j / nct) * TM;
        n0 = (j % nct) * TN;
        return j < rbn * nct;
    };
    auto next_valid = [&](int j, int& m0, int& n0) {
        while (j < jmax && !decode(j, m0, n0)) j += jstep;
        return j;
    };
    // ---- this workgroup's work list.  The XCD's valid tiles are j = 0 .. T - 1 (a prefix of the virtual tiles when the row count
    //      is the host's); its W workgroups take them round robin: `full` whole rounds, then `rem` tiles are left over.  With a
    //      workspace and rem <= W / 2 the LAST, partial round is not run as `rem` whole tiles on `rem` of the W compute units
    //      (10 000 x 1024 on 256 CUs: 320 tiles = 1.25 rounds cost two rounds of time, 38 %): each left-over tile's K range is cut
    //      into f = W / rem parts for f workgroups, partial sums go to the workspace and a small second launch adds them in part
    //      order and runs the epilogue (gemm_bf16_fixup_kernel; deterministic, no counters, nobody waits).
    const int W = jstep;
    int T, full, rem, f;
    mg_gemm_split(xcd, nrb, nct, rps, jmax, W, nk, T, full, rem, f);
    if (m_dev) {                                                   // device row count: the column-split map can have holes -- the
        full = 0;                                                  // plain round robin over the valid tiles, no K split
        rem = 0;
        f = 0;
    } else if (!ws_part) {
        f = 0;
    }
    // item i of this workgroup: i < full: tile jj0 + i W, all of K; then at most one more: a left-over tile (whole, f == 0) or
    // one K part of it
    const bool part_on = f > 0 && jj0 < rem * f;
    const int part_id = part_on ? jj0 % f : 0, part_tile = full * W + (part_on ? jj0 / f : 0);
    const int part_k0 = part_on ? part_id * nk / f : 0, part_k1 = part_on ? (part_id + 1) * nk / f : nk;
    int nitems, S;
    if (m_dev) {
        int m0, n0;
        nitems = 0;
        for (int j = jj0; j < jmax; j += jstep) nitems += decode(j, m0, n0) ? 1 : 0;
        S = nitems * nk;
    } else {
        const bool whole_rem = f == 0 && jj0 < rem;
        nitems = full + ((whole_rem || part_on) ? 1 : 0);
        S = full * nk + (whole_rem ? nk : 0) + (part_on ? part_k1 - part_k0 : 0);
    }
    if (nitems == 0) return;
    // item -> (tile, first k slice, slices)
    auto item = [&](int i, int& jcur, int& m0, int& n0, int& k0, int& kn) {
        if (m_dev) {                                               // walk the valid tiles of jj0, jj0 + W, ...
            jcur = next_valid(i == 0 ? jj0 : jcur + jstep, m0, n0);
            k0 = 0;
            kn = nk;
            return;
        }
        if (i < full) {
            jcur = jj0 + i * W;
            k0 = 0;
            kn = nk;
        } else if (part_on) {
            jcur = part_tile;
            k0 = part_k0;
            kn = part_k1 - part_k0;
        } else {
            jcur = full * W + jj0;
            k0 = 0;
            kn = nk;
        }
        decode(jcur, m0, n0);
    };

    if (wave >= 8) {
        // ---- producer q: A pieces q + 4 i, Bt pieces 32 + q + 4 i of every slice.  Piece p covers 8 rows x 128 B; lane
        //      (row_in = lane >> 3, slot = lane & 7) fetches the chunk that belongs in its slot: slot = chunk ^ ((row >> 1) & 7),
        //      and a piece starts at a multiple of 8 rows, so (row >> 1) & 7 = 4 (p & 1) + (row_in >> 1), p & 1 == q & 1.
        const int q = wave - 8;
        const int row_in = lane >> 3, slot = lane & 7;
        const int chunk = slot ^ (4 * (q & 1) + (row_in >> 1));
        int im0 = 0, in0 = 0, ik0 = 0, ikn = 0, ikt = 0, ig = 0, ii = 0, ij = 0;
        item(0, ij, im0, in0, ik0, ikn);
        const unsigned short* zsrc = reinterpret_cast<const unsigned short*>(g_zero16);
        auto issue = [&]() {                                       // slice ig of the stream -> stage ig % NSTAGE; then advance
            unsigned char* sb = smem + (size_t)(ig % NSTAGE) * STAGE_BYTES;
            const bool live = ig < S;                              // past the end: dummy pieces keep the operation count fixed
            const size_t koff = (size_t)(ik0 + ikt) * BK + chunk * 8;
            auto dma = [&](const unsigned short* src, int p) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(uintptr_t)(sb + (size_t)p * 1024), 16, 0, 0);
            };
#pragma unroll
            for (int i = 0; i < APP; ++i) {
                int row = im0 + (q + NPROD * i) * 8 + row_in;
                row = row < M ? row : M - 1;                       // rows beyond M: any valid row (never stored)
                dma(live ? A + (size_t)row * Kp + koff : zsrc, q + NPROD * i);
            }
#pragma unroll
            for (int i = 0; i < BPP; ++i) {
                int row = in0 + (q + NPROD * i) * 8 + row_in;
                row = row < N ? row : N - 1;
                dma(live ? Bt + (size_t)row * Kp + koff : zsrc, A_PIECES + q + NPROD * i);
            }
            ++ig;
            if (++ikt == ikn && ig < S) {
                ikt = 0;
                item(++ii, ij, im0, in0, ik0, ikn);
            }
        };
        // bare s_waitcnt + s_barrier: __syncthreads() carries vmcnt(0) and would wait for the slices just requested
        issue();
        issue();
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PPP) : "memory");      // slice 0 landed
        issue();
        for (int g = 0; g < S; ++g) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PPP) : "memory");  // slice g+1 landed; the stage of slice g is free
            issue();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // dummy pieces must not outlive the workgroup's LDS
        return;
    }

    // ---- compute wave: tile rows wr*64.., columns wc*64.. (4 x 4 MFMA tiles)
    const int wr = wave >> 1, wc = wave & 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses: lane (r = lane & 15, g = lane >> 4) reads chunk (4 s + g) ^ ((r >> 1) & 7) of row (tile * 16 + r);
    // tile i of a wave lies i * 2 KiB further on (immediate offset).  The reads go through mg_lds_read128 (inline asm):
    // as ordinary ds_reads hipcc guards each group with s_waitcnt vmcnt(0) against the LDS-DMA in flight.
    const int fr = lane & 15, fg = lane >> 4;
    const unsigned lds0 = mg_lds_addr(smem);
    unsigned aoff[2], boff[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        aoff[s2] = lds0 + ((wr * 64 + fr) * 8 + ((4 * s2 + fg) ^ ((fr >> 1) & 7))) * 16;
        boff[s2] = lds0 + A_BYTES + ((wc * 64 + fr) * 8 + ((4 * s2 + fg) ^ ((fr >> 1) & 7))) * 16;
    }
    u32x4 a[2][4], b[2][4];
    auto reads = [&](int stage, int s, int buf) {
        const unsigned so = (unsigned)stage * STAGE_BYTES;
        a[buf][0] = mg_lds_read128<0>(aoff[s] + so);
        a[buf][1] = mg_lds_read128<2048>(aoff[s] + so);
        a[buf][2] = mg_lds_read128<4096>(aoff[s] + so);
        a[buf][3] = mg_lds_read128<6144>(aoff[s] + so);
        b[buf][0] = mg_lds_read128<0>(boff[s] + so);
        b[buf][1] = mg_lds_read128<2048>(boff[s] + so);
        b[buf][2] = mg_lds_read128<4096>(boff[s] + so);
        b[buf][3] = mg_lds_read128<6144>(boff[s] + so);
    };
    // tiles are computed transposed (A operand = Bt fragment) so a lane ends up with four consecutive C columns
    auto mmas = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[buf][jj]),
                                                                    __builtin_bit_cast(bf16x8, a[buf][i]), acc[i][jj], 0, 0, 0);
    };
    int cm0 = 0, cn0 = 0, ck0 = 0, ckn = 0, ckt = 0, ci = 0, cj = 0;
    item(0, cj, cm0, cn0, ck0, ckn);
    asm volatile("s_barrier" ::: "memory");                        // slice 0 landed (the producers waited for it)
    reads(0, 0, 0);
    // software pipeline: the reads of k-step 1 are issued before the MFMAs of k-step 0 and -- behind the slice barrier in the
    // MIDDLE of the iteration -- the reads of the next slice's k-step 0 before the MFMAs of k-step 1
    for (int g = 0; g < S; ++g) {
        reads(g % NSTAGE, 1, 1);
        mg_lds_wait<8>();                                          // k-step 0 landed (the 8 reads of k-step 1 are behind it)
        __builtin_amdgcn_sched_barrier(0);
        mmas(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // done with the stage of slice g; slice g+1 landed
        reads((g + 1) % NSTAGE, 0, 0);
        mg_lds_wait<8>();
        __builtin_amdgcn_sched_barrier(0);
        mmas(1);
        __builtin_amdgcn_sched_barrier(0);
        if (++ckt < ckn) continue;
        ckt = 0;
        if (part_on && ci == full) {
            // ---- a K part of a left-over tile: the partial sums go to this workgroup's slot of the workspace
            //      ([slot][wave][i][jj][lane] x 16 B); gemm_bf16_fixup_kernel, the next launch on the stream, adds a tile's f
            //      partials in part order and runs the epilogue
            float* wp = ws_part + (size_t)(xcd * W + jj0) * (TM * TN) + ((size_t)(wave * 16) * 64 + lane) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    __builtin_nontemporal_store(acc[i][jj], reinterpret_cast<f32x4*>(wp + (size_t)(i * 4 + jj) * 64 * 4));
            return;
        }
        // ---- epilogue: acc[i][jj][r] = C[cm0 + wr*64 + 16 i + (lane & 15)][cn0 + wc*64 + 16 jj + 4 (lane >> 4) + r]
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int n = cn0 + wc * 64 + jj * 16 + fg * 4;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (bias && n < N) bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = cm0 + wr * 64 + i * 16 + fr;
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = mg_act(acc[i][jj][r] + bv[r], act);
                acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (m < M && n < N) {
                    if (c_bf16) {                                  // C is a bf16 matrix (ldc in elements): the operand of the next product
                        unsigned lo, hi;
                        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(o[0]), "v"(o[1]));
                        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(o[2]), "v"(o[3]));
                        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<u32x2_t*>(reinterpret_cast<unsigned short*>(C) + (size_t)m * ldc + n) = u32x2_t{lo, hi};
                    } else {
                        *reinterpret_cast<f32x4*>(C + (size_t)m * ldc + n) = o;
                    }
                }
            }
        }
        if (g + 1 < S) item(++ci, cj, cm0, cn0, ck0, ckn);
    }
}

// The left-over tiles of gemm_bf16_nt_kernel's last round: workgroup (xcd, r) adds the f partial sums of left-over tile r of XCD
// xcd in part order and runs that kernel's epilogue.  512 threads; thread t owns the accumulator slots of compute wave t >> 6
// exactly as that wave wrote them.
__global__ __launch_bounds__(512) void gemm_bf16_fixup_kernel(const float* __restrict__ ws_part, int M, int N, const float* __restrict__ bias,
                                                              float* __restrict__ C, int ldc, int act, int nrb, int nct, int rps,
                                                              int jmax, int W, int nk, int c_bf16) {
    const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
    int T, full, rem, f;
    mg_gemm_split(xcd, nrb, nct, rps, jmax, W, nk, T, full, rem, f);
    if (f == 0 || r >= rem) return;
    const int j = full * W + r;
    int m0, n0;
    if (rps == 0) {
        const int cpx = jmax / nrb;
        m0 = (j % nrb) * TM;
        n0 = (xcd * cpx + j / nrb) * TN;
    } else {
        m0 = (xcd * nrb / 8 + j / nct) * TM;
        n0 = (j % nct) * TN;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;        // the compute wave whose slots this thread sums
    const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fg = lane >> 4;
    const float* base = ws_part + (size_t)(xcd * W + r * f) * (TM * TN) + ((size_t)(wave * 16) * 64 + lane) * 4;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int n = n0 + wc * 64 + jj * 16 + fg * 4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias && n < N) bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wr * 64 + i * 16 + fr;
            f32x4 sum = {0.f, 0.f, 0.f, 0.f};
            for (int pt = 0; pt < f; ++pt)
                sum += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + (size_t)pt * (TM * TN) + (size_t)(i * 4 + jj) * 64 * 4));
            f32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = mg_act(sum[q] + bv[q], act);
            if (m < M && n < N) {
                if (c_bf16) {
                    unsigned lo, hi;
                    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(o[0]), "v"(o[1]));
                    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(o[2]), "v"(o[3]));
                    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<u32x2_t*>(reinterpret_cast<unsigned short*>(C) + (size_t)m * ldc + n) = u32x2_t{lo, hi};
                } else {
                    *reinterpret_cast<f32x4*>(C + (size_t)m * ldc + n) = o;
                }
            }
        }
    }
}

}  // namespace

extern "C" int mgnns_transpose_cast_bf16(const float* x, int rows, int cols, int ld, void* y, mgnns_stream_t stream) {
    MG_REQUIRE(x && y, "mgnns_transpose_cast_bf16: null pointer");
    MG_REQUIRE(rows > 0 && cols > 0 && ld >= rows, "mgnns_transpose_cast_bf16: bad dims rows=%d cols=%d ld=%d", rows, cols, ld);
    dim3 grid((cols + 31) / 32, (ld + 31) / 32);
    hipLaunchKernelGGL(transpose_cast_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld,
                       reinterpret_cast<unsigned short*>(y));
    MG_CHECK_LAUNCH("mgnns_transpose_cast_bf16");
    return 0;
}

// ---- 256 x 256 tiles (round 4) -----------------------------------------------------------------------------------------------
// The 256 x 128 kernel above is bound by what the XCD's L2 delivers, not by the matrix pipe: 10 000 x 1024 x 10 000 moves 2.5 GB
// from L2 to LDS (TCC hits + misses: 20.1 M lines per launch) for 532 MB of HBM reads and 275 us = 9.3 TB/s over the eight L2s,
// while its MFMAs are 35 % of the time.  A 256 x 256 tile needs 2 / 256 operand bytes per MAC instead of 3 / 256: a third less L2
// traffic for the same product -- a tile's 157 slices take 270 us with every CU busy, against 200 us for half as many MACs.
// 512 threads, eight waves = two per SIMD (the 128 accumulator registers of a 64 x 128 wave tile leave no room for a third wave,
// i.e. for dedicated producer waves: every wave requests its share of the DMA pieces of a slice between the MFMAs of an earlier
// one).
// Work list of a workgroup, as in the kernel above: whole tiles round robin over the XCD's W workgroups (tile jj0 + i W: the
// workgroups of a round sit on neighbouring tiles and march along K together, which is what keeps their operand slices in the
// XCD's L2 -- dealing the tile stream out in equal runs of SLICES balanced every workgroup to the slice but put neighbours at
// different k: 1.6 GB from HBM instead of 0.5, 304 us instead of 270 for 10 000 x 1024), then the rem = T mod W left-over tiles cut
// along K into f = W / rem parts each; partial sums go to the workgroup's slot of the workspace and gemm_bf16_fixup_256_kernel,
// the next launch on the stream, adds a tile's parts in K order and runs the epilogue: deterministic, no counters, nobody waits.
namespace {
constexpr int TM2 = 256, TN2 = 256;

// the epilogue of a 64 x 128 wave tile: acc[i][jj][r] = C[cm0 + wr*64 + 16 i + (lane & 15)][cn0 + wc*128 + 16 jj + 4 (lane >> 4) + r]
__device__ __forceinline__ void gemm256_store(f32x4 (&acc)[4][8], int cm0, int cn0, int wr, int wc, int fr, int fg, int M, int N,
                                              const float* __restrict__ bias, float* __restrict__ C, int ldc, int act, int c_bf16) {
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int n = cn0 + wc * 128 + jj * 16 + fg * 4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias && n < N) bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = cm0 + wr * 64 + i * 16 + fr;
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = mg_act(acc[i][jj][r] + bv[r], act);
            if (m < M && n < N) {
                if (c_bf16) {                                      // C is a bf16 matrix (ldc in elements): the operand of the next product
                    unsigned lo, hi;
                    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(o[0]), "v"(o[1]));
                    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(o[2]), "v"(o[3]));
                    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<u32x2_t*>(reinterpret_cast<unsigned short*>(C) + (size_t)m * ldc + n) = u32x2_t{lo, hi};
                } else {
                    *reinterpret_cast<f32x4*>(C + (size_t)m * ldc + n) = o;
                }
            }
        }
    }
}

// (full, rem, f) of an XCD with T tiles and W workgroups: `full` rounds of whole tiles, then rem tiles in f K parts each
__host__ __device__ inline void gemm256_split(int T, int W, int nk, int& full, int& rem, int& f) {
    full = T / W;
    rem = T - full * W;
    f = rem > 0 ? W / rem : 0;
    if (f > 8) f = 8;
    if (f < 2 || f * 4 > nk) f = rem > 0 ? 1 : 0;                  // (f == 1: the left-over tiles stay whole)
}

// The tile stream runs through a FOUR-stage ring of 32-wide K slices (32 KB each): three slices in flight.  (First form: two
// 64-wide stages, the next slice requested while the one before is multiplied -- one L2 round trip per slice; 427 us against 422
// for 10 000 x 2048 x 10 000 on one box.  The difference is small because neither is what a slice costs, see mmas below.)
// Operand rows are 64 B here: [row][4 chunks of 16 B], chunk c at slot c ^ g[(row >> 2) & 3], g = {2, 0, 1, 3} (the image bank's
// operand image: conflict-free ds_read_b128 fragments); a DMA piece is 16 rows x 64 B.
#ifdef MG_GEMM_TRACE
__device__ unsigned long long g_gemm_trace[4];                     // profiling aid: s_memtime ticks of workgroup 0 / 100 (wave 0)
#endif
constexpr int BK3 = 32, NST3 = 4, OPB3 = TM2 * BK3 * 2, STG3 = 2 * OPB3;           // 16 KB per operand, 32 KB per stage
constexpr size_t SMEM3_BYTES = (size_t)NST3 * STG3;
__global__ __launch_bounds__(512) void gemm_bf16_nt_256_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ Bt,
                                                                int M, int N, int Kp, const float* __restrict__ bias,
                                                                float* __restrict__ C, int ldc, int act, int nrb, int nct, int c_bf16,
                                                                float* __restrict__ ws_part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef MG_GEMM_TRACE
    const unsigned long long tr0 = __builtin_amdgcn_s_memtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, jj0 = blockIdx.x >> 3, W = gridDim.x >> 3;
    const int nk = Kp / BK;                                        // (the work list counts K in 64-wide slices, as the fix-up launch does)
    const int rb0 = xcd * nrb / 8, T = ((xcd + 1) * nrb / 8 - rb0) * nct;
    int full, rem, f;
    gemm256_split(T, W, nk, full, rem, f);
    const bool last_on = f >= 1 && jj0 < rem * f;
    const int nitem = full + (last_on ? 1 : 0);
    if (nitem == 0) return;
    const int l_tile = full * W + (last_on ? jj0 / f : 0), l_part = last_on ? jj0 % f : 0;
    const int l_k0 = last_on ? l_part * nk / f : 0, l_k1 = last_on ? (l_part + 1) * nk / f : nk;
    const int S = 2 * (full * nk + (last_on ? l_k1 - l_k0 : 0));   // 32-wide slices of this workgroup
    auto item = [&](int i, int& j, int& k0, int& k1) {             // tile, first and last + 1 32-wide slice (wave-uniform)
        const bool whole = i < full;
        j = __builtin_amdgcn_readfirstlane(whole ? jj0 + i * W : l_tile);
        k0 = __builtin_amdgcn_readfirstlane(whole ? 0 : 2 * l_k0);
        k1 = __builtin_amdgcn_readfirstlane(whole ? 2 * nk : 2 * l_k1);
    };
    // ---- requests: piece p of an operand tile covers 16 rows x 64 B; lane (row_in = lane >> 2, slot = lane & 3) fetches the chunk
    //      that belongs in its slot; wave w takes A pieces w, w + 8 and Bt pieces w, w + 8 of every slice
    const int row_in = lane >> 2, slot = lane & 3;
    const int chunk = slot ^ ((0xD2 >> (2 * ((row_in >> 2) & 3))) & 3);
    const unsigned voff = (unsigned)((row_in * Kp + chunk * 8) * 2);
    int ii = 0, it, ik, ik1;
    item(0, it, ik, ik1);
    __amdgpu_buffer_rsrc_t ra, rb;
    auto open_tile = [&](int j) {
        const int m0 = (rb0 + j / nct) * TM2, n0 = (j % nct) * TN2;
        const int mr = M - m0 < TM2 ? M - m0 : TM2, nr = N - n0 < TN2 ? N - n0 : TN2;
        ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A + (size_t)m0 * Kp), 0, mr * Kp * 2, 0x00027000);
        rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Bt + (size_t)n0 * Kp), 0, nr * Kp * 2, 0x00027000);
    };
    open_tile(it);
    auto issue_one = [&](int g, int i) {                           // request i (0..3) of slice g
        unsigned char* sb = smem + (size_t)(g & (NST3 - 1)) * STG3;
        const int p = wave + 8 * (i & 1);
        const unsigned soff = (unsigned)((p * 16 * Kp + ik * BK3) * 2);
        if (i < 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(uintptr_t)(sb + (size_t)p * 1024), 16, voff, soff, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(uintptr_t)(sb + OPB3 + (size_t)p * 1024), 16, voff, soff, 0, 0);
    };
    auto issue_done = [&]() {
        if (++ik == ik1 && ++ii < nitem) {
            item(ii, it, ik, ik1);
            open_tile(it);
        }
    };
    // ---- compute: wave tile rows wr * 64 .., columns wc * 128 .. (4 x 8 MFMA tiles)
    const int wr = wave >> 1, wc = wave & 1;
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    const unsigned lds0 = mg_lds_addr(smem);
    const unsigned swz = (unsigned)((fg ^ ((0xD2 >> (2 * ((fr >> 2) & 3))) & 3)) << 4);
    const unsigned aoff = lds0 + (unsigned)((wr * 64 + fr) * 64) + swz, boff = lds0 + OPB3 + (unsigned)((wc * 128 + fr) * 64) + swz;
    u32x4 a[4], b[8];
    auto reads = [&](int stage) {
        const unsigned so = (unsigned)stage * STG3;
        a[0] = mg_lds_read128<0>(aoff + so);
        a[1] = mg_lds_read128<1024>(aoff + so);
        a[2] = mg_lds_read128<2048>(aoff + so);
        a[3] = mg_lds_read128<3072>(aoff + so);
        b[0] = mg_lds_read128<0>(boff + so);
        b[1] = mg_lds_read128<1024>(boff + so);
        b[2] = mg_lds_read128<2048>(boff + so);
        b[3] = mg_lds_read128<3072>(boff + so);
        b[4] = mg_lds_read128<4096>(boff + so);
        b[5] = mg_lds_read128<5120>(boff + so);
        b[6] = mg_lds_read128<6144>(boff + so);
        b[7] = mg_lds_read128<7168>(boff + so);
    };
    // tiles are computed transposed (A operand = Bt fragment) so a lane ends up with four consecutive C columns; half h = column
    // tiles 4 h .. 4 h + 3; two of the four requests of slice g + 3 ride on each half, behind every eighth MFMA.  (Fragments of
    // the NEXT slice read under these MFMAs -- 96 more registers, one slice less in flight -- measured no faster: a slice costs
    // ~1850 cycles for 1024 of MFMAs either way; what the waves lose is the ISSUE of their 4 + 4 requests per 64 k, 100-185 cycles
    // each next to fragment reads, and 128 accumulator registers leave no room for producer waves.)
    auto mmas = [&](int h, int g, bool req) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                const int jj = 4 * h + j4;
                acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[jj]), __builtin_bit_cast(bf16x8, a[i]),
                                                                    acc[i][jj], 0, 0, 0);
                if (req && j4 == 3 && (i & 1)) issue_one(g + NST3 - 1, 2 * h + (i >> 1));
            }
    };
    for (int g = 0; g < NST3 - 1; ++g) {                           // (S >= 16: a part is at least eight 64-wide slices)
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_one(g, i);
        issue_done();
    }
    asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");  // slice 0 landed (everybody's pieces)
    int ci = 0, ct, ck, ck1;
    item(0, ct, ck, ck1);
    for (int g = 0; g < S; ++g) {
        const bool req = g + NST3 - 1 < S;                         // slice g + 3 -> the stage of slice g - 1: read by everybody before the last barrier
        reads(g & (NST3 - 1));
        mg_lds_wait<4>();                                          // a[0..3], b[0..3]
        __builtin_amdgcn_sched_barrier(0);
        mmas(0, g, req);
        __builtin_amdgcn_sched_barrier(0);
        mg_lds_wait<0>();
        __builtin_amdgcn_sched_barrier(0);
        mmas(1, g, req);
        if (req) issue_done();
        __builtin_amdgcn_sched_barrier(0);
        // slice g + 1 landed (its own pieces: at most the 8 of slices g + 2, g + 3 are younger); everybody is done with slice g
        if (req) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        else if (g + 2 < S) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (++ck < ck1) continue;
        // ---- item ci ends here
        const int cm0 = (rb0 + ct / nct) * TM2, cn0 = (ct % nct) * TN2;
        if (ci < full || f == 1) {
            gemm256_store(acc, cm0, cn0, wr, wc, fr, fg, M, N, bias, C, ldc, act, c_bf16);
        } else {
            float* wp = ws_part + (size_t)(xcd * W + jj0) * (TM2 * TN2) + ((size_t)(wave * 32) * 64 + lane) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    __builtin_nontemporal_store(acc[i][jj], reinterpret_cast<f32x4*>(wp + (size_t)(i * 8 + jj) * 64 * 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (++ci < nitem) item(ci, ct, ck, ck1);
    }
#ifdef MG_GEMM_TRACE
    if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) {
        g_gemm_trace[blockIdx.x ? 2 : 0] = __builtin_amdgcn_s_memtime() - tr0;
        g_gemm_trace[blockIdx.x ? 3 : 1] = (unsigned long long)S;
    }
#endif
}

// Left-over tile r of XCD blockIdx.x & 7 (r = blockIdx.x >> 3): add its f parts in K order (= workgroup order) and run the
// epilogue.  512 threads; thread t owns the accumulator slots of compute wave t >> 6 exactly as that wave wrote them.
__global__ __launch_bounds__(512) void gemm_bf16_fixup_256_kernel(const float* __restrict__ ws_part, int M, int N, int Kp,
                                                                  const float* __restrict__ bias, float* __restrict__ C, int ldc, int act,
                                                                  int nrb, int nct, int W, int c_bf16) {
    const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
    const int nk = Kp / BK;
    const int rb0 = xcd * nrb / 8, T = ((xcd + 1) * nrb / 8 - rb0) * nct;
    int full, rem, f;
    gemm256_split(T, W, nk, full, rem, f);
    if (f < 2 || r >= rem) return;
    const int t = full * W + r;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < f; ++p) {
        const float* wp = ws_part + (size_t)(xcd * W + r * f + p) * (TM2 * TN2) + ((size_t)(wave * 32) * 64 + lane) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp + (size_t)(i * 8 + jj) * 64 * 4));
                acc[i][jj] += v;
            }
    }
    gemm256_store(acc, (rb0 + t / nct) * TM2, (t % nct) * TN2, wave >> 1, wave & 1, lane & 15, lane >> 4, M, N, bias, C, ldc, act, c_bf16);
}

}  // namespace

namespace {
template <int N_> struct mg_ic { static constexpr int v = N_; };
template <typename F> __device__ __forceinline__ void mg_mha_static_for_2(F&& f) { f(mg_ic<0>{}); f(mg_ic<1>{}); }
}  // namespace

// ---- 160 x 256 tiles (round 5) -----------------------------------------------------------------------------------------------
// What the two kernels above leave on the table at 10 000 x 1024 x 10 000: the 256 x 128 kernel streams at what the L2s deliver
// (9.3 TB/s) but its 320 tiles are 1.25 rounds of 256 workgroups (a K split of the last quarter round + a fix-up launch: 254-265 us);
// the 256 x 256 kernel has a third less traffic but 160 tiles on 256 CUs and no room for producer waves.  A 160 x 256 tile: 63 row
// blocks x 4 column tiles = 252 tiles = ONE round on 252 of the 256 CUs, nothing to split or fix up; (160 + 256) operand rows per
// 40 960 MACs = 13 % less L2 -> LDS traffic per MAC than 256 x 128 (2.10 GB instead of 2.46); and its 80 x 64 wave tile needs 80
// accumulator registers, which still leaves room for the third wave per SIMD, i.e. for the four producer waves that make the
// 256 x 128 kernel reach the L2 rate.  32-wide K slices (26 KB: 10 + 16 DMA pieces of 16 rows x 64 B) through a FIVE-stage ring:
// four slices = 104 KB in flight per CU (256 x 128: two of 48 KB).  Operand rows in LDS as in the 256 x 256 kernel (64 B, chunk c at
// slot c ^ g[(row >> 2) & 3]).  Whole tiles only, round robin over the XCD's workgroups in the XCD-aware order of the kernels above.
namespace {
#ifndef MG_GEMM160_NST
#define MG_GEMM160_NST 5                                             // stages of the slice ring (measured: 4: same, 3: 239 us against 226)
#endif
constexpr int TM4 = 160, TN4 = 256, BK4 = 32, NST4 = MG_GEMM160_NST;
constexpr int A4_BYTES = TM4 * BK4 * 2, B4_BYTES = TN4 * BK4 * 2, STG4 = A4_BYTES + B4_BYTES;          // 10 KB + 16 KB
constexpr int A4_PIECES = A4_BYTES / 1024, B4_PIECES = B4_BYTES / 1024;                                  // 10 + 16
constexpr int PPP4 = (A4_PIECES + B4_PIECES + NPROD - 1) / NPROD;                                       // 7 requests per producer and slice (two of the 28 are dummies)
constexpr size_t SMEM4_BYTES = (size_t)NST4 * STG4 + 1024;                                              // + the dummies' landing strip
constexpr int NTHR4 = 768;

__global__ __launch_bounds__(NTHR4) void gemm_bf16_nt_160_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ Bt,
                                                                 int M, int N, int Kp, const float* __restrict__ bias,
                                                                 float* __restrict__ C, int ldc, int act, int nrb, int nct, int c_bf16) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, jj0 = blockIdx.x >> 3, W = gridDim.x >> 3;
    const int nk = Kp / BK4;                                       // 32-wide slices of a tile
    const int rb0 = xcd * nrb / 8, T = ((xcd + 1) * nrb / 8 - rb0) * nct;
    const int nitem = jj0 < T ? (T - jj0 + W - 1) / W : 0;         // tiles jj0, jj0 + W, ...
    if (nitem == 0) return;
    const int S = nitem * nk;

    if (wave >= 8) {
        // ---- producer q: requests q, q + 4, .. q + 24 of every slice: request p < 10 = A piece p, p < 26 = Bt piece p - 10, else a
        //      dummy (16 B of zeros per lane onto the landing strip: every producer issues the SAME number of requests per slice, which
        //      is what lets it wait with a constant vmcnt).  A piece = 16 rows x 64 B; lane (row_in = lane >> 2, slot = lane & 3)
        //      fetches the chunk that belongs in its slot.
        const int q = wave - 8;
        const int row_in = lane >> 2, slot = lane & 3;
        const int chunk = slot ^ ((0xD2 >> (2 * ((row_in >> 2) & 3))) & 3);
        const unsigned short* zsrc = reinterpret_cast<const unsigned short*>(g_zero16);
        unsigned char* strip = smem + (size_t)NST4 * STG4;
        int ig = 0, ikt = 0, ii = 0, im0, in0;
        auto open = [&](int i) {
            const int j = jj0 + i * W;
            im0 = (rb0 + j / nct) * TM4;
            in0 = (j % nct) * TN4;
        };
        open(0);
        auto issue = [&]() {                                       // slice ig of the stream -> stage ig % NST4; then advance
            unsigned char* sb = smem + (size_t)(ig % NST4) * STG4;
            const bool live = ig < S;                              // past the end: dummies keep the request count fixed
            const size_t koff = (size_t)ikt * BK4 + chunk * 8;
#pragma unroll
            for (int i = 0; i < PPP4; ++i) {
                const int p = q + NPROD * i;                       // (wave-uniform)
                const unsigned short* src = zsrc;
                unsigned char* dst = strip;
                if (live && p < A4_PIECES) {
                    int row = im0 + p * 16 + row_in;
                    row = row < M ? row : M - 1;                   // rows beyond M: any valid row (never stored)
                    src = A + (size_t)row * Kp + koff;
                    dst = sb + (size_t)p * 1024;
                } else if (live && p < A4_PIECES + B4_PIECES) {
                    int row = in0 + (p - A4_PIECES) * 16 + row_in;
                    row = row < N ? row : N - 1;
                    src = Bt + (size_t)row * Kp + koff;
                    dst = sb + A4_BYTES + (size_t)(p - A4_PIECES) * 1024;
                }
#ifndef MG_GEMM160_AUX
#define MG_GEMM160_AUX 0                                           // cache policy bits of the operand requests (measured: nt 359 us against 226, sc0 / sc1 no change)
#endif
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16, 0, MG_GEMM160_AUX);
            }
            ++ig;
            if (++ikt == nk && ig < S) {
                ikt = 0;
                open(++ii);
            }
        };
        // bare s_waitcnt + s_barrier: __syncthreads() carries vmcnt(0) and would wait for the slices just requested
#pragma unroll
        for (int g = 0; g < NST4 - 1; ++g) issue();
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NST4 - 2) * PPP4) : "memory");     // slice 0 landed
        issue();
        for (int g = 0; g < S; ++g) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NST4 - 2) * PPP4) : "memory"); // slice g + 1 landed; the stage of slice g is free
            issue();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // nothing may land after the workgroup's LDS is gone
        return;
    }

    // ---- compute wave: tile rows wr * 80 .., columns wc * 64 .. (5 x 4 MFMA tiles), one k-step per slice
    const int wr = wave >> 2, wc = wave & 3;
    f32x4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    const unsigned lds0 = mg_lds_addr(smem);
    const unsigned swz = (unsigned)((fg ^ ((0xD2 >> (2 * ((fr >> 2) & 3))) & 3)) << 4);
    const unsigned aoff = lds0 + (unsigned)((wr * 80 + fr) * 64) + swz;
    const int bdelta = A4_BYTES + (wc * 64 - wr * 80) * 64;         // Bt fragment row of this lane - its A fragment row: wave-uniform (an SGPR, not a second address register)
    u32x4 a[2][5], b[2][4];
    auto reads = [&](int stage, int buf) {
        const unsigned so = (unsigned)stage * STG4;
        const unsigned ao = aoff + so, bo = ao + (unsigned)bdelta;
        a[buf][0] = mg_lds_read128<0>(ao);
        a[buf][1] = mg_lds_read128<1024>(ao);
        a[buf][2] = mg_lds_read128<2048>(ao);
        a[buf][3] = mg_lds_read128<3072>(ao);
        a[buf][4] = mg_lds_read128<4096>(ao);
        b[buf][0] = mg_lds_read128<0>(bo);
        b[buf][1] = mg_lds_read128<1024>(bo);
        b[buf][2] = mg_lds_read128<2048>(bo);
        b[buf][3] = mg_lds_read128<3072>(bo);
    };
    // tiles are computed transposed (A operand = Bt fragment) so a lane ends up with four consecutive C columns
    auto mmas = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[buf][jj]),
                                                                    __builtin_bit_cast(bf16x8, a[buf][i]), acc[i][jj], 0, 0, 0);
    };
    int ci = 0, ckt = 0;
    // ---- L2 prefetch by the COMPUTE waves (round 6; MG_GEMM160_PF slices ahead, 0 = off) -----------------------------------------
    // The operand stream is bound by the round trip of a slice's FRESH lines (NOTES_r06 2a): the producers' requests complete in order,
    // so a miss far ahead would hold back the nearer slices if a producer issued it -- the compute waves have no vector-memory
    // traffic of their own, so their vmcnt is free: every second slice each of them touches one 128-B line per lane (8 waves x 64
    // lanes >= the 160 + 256 operand rows of a slice pair) MG_GEMM160_PF slices ahead, as a 4-byte LDS-DMA onto the landing strip
    // (no destination register, nobody waits for it).  The producers' requests then find their lines in the L2.
#ifndef MG_GEMM160_PF
#define MG_GEMM160_PF 0
#endif
    constexpr int PF = MG_GEMM160_PF;
    // waves 0-2 touch the tile's 160 A rows (lane = row, 32 lanes idle), waves 3-6 its 256 Bt rows, wave 7 nothing: the base is wave-uniform
    // (a buffer resource in SGPRs, rows beyond the matrix fall outside its bounds: no request), the lane keeps ONE register: its row offset
    const bool pf_a = wave < 3, pf_on = wave < 7;
    const int pf_local = pf_a ? wave * 64 + lane : (wave - 3) * 64 + lane;
    const unsigned pf_voff = (unsigned)pf_local * (unsigned)Kp * 2u;
    __amdgpu_buffer_rsrc_t pf_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A), 0, 0, 0x00027000);
    auto pf_open = [&](int i) {
        const int j = jj0 + i * W;
        const int m0 = (rb0 + j / nct) * TM4, n0 = (j % nct) * TN4;
        const int lim = pf_a ? (M - m0 < TM4 ? M - m0 : TM4) : (N - n0 < TN4 ? N - n0 : TN4);
        const unsigned short* base = pf_a ? A + (size_t)m0 * Kp : Bt + (size_t)n0 * Kp;
        pf_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(base), 0, pf_on && lim > 0 ? lim * Kp * 2 : 0, 0x00027000);
    };
    if (PF > 0) pf_open(0);
    asm volatile("s_barrier" ::: "memory");                        // slice 0 landed (the producers waited for it)
    reads(0, 0);
    // per slice: its fragments are in registers -> barrier (everybody is done with the slice's stage; the next slice landed) -> the
    // next slice's reads go out -> this slice's 20 MFMAs run over them.  Two slices per trip: the fragment buffers are named, not indexed.
    auto slice = [&](int g, int buf) {
        mg_lds_wait<0>();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
#ifndef MG_GEMM160_ABLATE
#define MG_GEMM160_ABLATE 0      // measurement builds only (results wrong on purpose): 1 = no MFMAs, 2 = no fragment reads either -- the operand stream alone
#endif
#if !(MG_GEMM160_ABLATE & 2)
        if (g + 1 < S) reads((g + 1) % NST4, buf ^ 1);
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (PF > 0 && (ckt & 1) == 0 && ckt + PF < nk)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(pf_rs, (__attribute__((address_space(3))) void*)(uintptr_t)(smem + (size_t)NST4 * STG4), 4, pf_voff,
                                                     (ckt + PF) * BK4 * 2, 0, 0);
#if !(MG_GEMM160_ABLATE & 1)
        mmas(buf);
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (++ckt < nk) return;
        ckt = 0;
        if (PF > 0 && ci + 1 < nitem) pf_open(ci + 1);
        // ---- epilogue: acc[i][jj][r] = C[cm0 + wr*80 + 16 i + (lane & 15)][cn0 + wc*64 + 16 jj + 4 (lane >> 4) + r]
        const int j = jj0 + ci * W;
        const int cm0 = (rb0 + j / nct) * TM4, cn0 = (j % nct) * TN4;
        ++ci;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int n = cn0 + wc * 64 + jj * 16 + fg * 4;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (bias && n < N) bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int m = cm0 + wr * 80 + i * 16 + fr;
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = mg_act(acc[i][jj][r] + bv[r], act);
                acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (m < M && m < cm0 + TM4 && n < N) {
                    if (c_bf16) {                                  // C is a bf16 matrix (ldc in elements): the operand of the next product
                        unsigned lo, hi;
                        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(o[0]), "v"(o[1]));
                        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(o[2]), "v"(o[3]));
                        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<u32x2_t*>(reinterpret_cast<unsigned short*>(C) + (size_t)m * ldc + n) = u32x2_t{lo, hi};
                    } else {
                        // every workgroup of the one round stores its 164 KB at the same time: past the L2 (222 against 225-228 us)
                        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(C + (size_t)m * ldc + n));
                    }
                }
            }
        }
    };
    for (int g = 0; g < S; g += 2) {
        slice(g, 0);
        if (g + 1 < S) slice(g + 1, 1);
    }
}

}  // namespace

// ---- 160 x 256 tiles, 64-wide K slices (round 6) ---------------------------------------------------------------------------------
// The L2 serves REQUESTS, not bytes (tools/dev/micro/l2_rowseg.hip: a 64-B row segment costs what a 128-B one costs; 33 G requests per
// second and XCD), and the kernel above asks for its operands as 64-B row segments (32-wide slices): 33 M requests per 10 000 x 1024 x
// 10 000 launch, 57 % of that capacity, and anything that adds requests costs time in proportion (the L2 prefetch: + 50 % requests,
// + 22 % time).  The same tile with 64-wide slices asks for whole 128-B lines: half the requests for the same bytes.  Three stages of 52 KB
// (two slices = 104 KB in flight, as before), pieces of 8 rows x 128 B, 13 requests per producer and slice (no dummies), operand rows in LDS
// as in the 256 x 128 kernel (chunk c of a row at slot c ^ ((row >> 1) & 7)), two k-steps per slice software-pipelined across the slice
// barrier.  MGNNS_GEMM160_BK=32 keeps the kernel above.
namespace {
constexpr int BK6 = 64, NST6 = 3;
constexpr int A6_BYTES = TM4 * BK6 * 2, B6_BYTES = TN4 * BK6 * 2, STG6 = A6_BYTES + B6_BYTES;          // 20 KB + 32 KB
constexpr int A6_PIECES = A6_BYTES / 1024, B6_PIECES = B6_BYTES / 1024;                                  // 20 + 32 = 52 = 4 producers x 13
constexpr int PPP6 = (A6_PIECES + B6_PIECES) / NPROD;
static_assert(PPP6 * NPROD == A6_PIECES + B6_PIECES, "every producer issues the same number of pieces");
constexpr size_t SMEM6_BYTES = (size_t)NST6 * STG6;
static_assert(SMEM6_BYTES <= 160 * 1024, "LDS");

__global__ __launch_bounds__(NTHR4) void gemm_bf16_nt_160k_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ Bt,
                                                                  int M, int N, int Kp, const float* __restrict__ bias,
                                                                  float* __restrict__ C, int ldc, int act, int nrb, int nct, int c_bf16) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];      // (the k-step XOR below relies on 128-B aligned rows)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, jj0 = blockIdx.x >> 3, W = gridDim.x >> 3;
    const int nk = Kp / BK6;                                       // 64-wide slices of a tile (Kp % 64 == 0: the launcher checks)
    const int rb0 = xcd * nrb / 8, T = ((xcd + 1) * nrb / 8 - rb0) * nct;
    const int nitem = jj0 < T ? (T - jj0 + W - 1) / W : 0;         // tiles jj0, jj0 + W, ...
    if (nitem == 0) return;
    const int S = nitem * nk;

    if (wave >= 8) {
        // ---- producer q: requests q, q + 4, .. q + 48 of every slice: request p < 20 = A piece p, else Bt piece p - 20.  A piece = 8 rows x
        //      128 B; lane (row_in = lane >> 3, slot = lane & 7) fetches the chunk that belongs in its slot: slot = chunk ^ ((row >> 1) & 7),
        //      a piece starts at a multiple of 8 rows, so (row >> 1) & 7 = 4 (piece & 1) + (row_in >> 1).
        const int q = wave - 8;
        const int row_in = lane >> 3, slot = lane & 7;
        int ig = 0, ikt = 0, ii = 0, im0, in0;
        auto open = [&](int i) {
            const int j = jj0 + i * W;
            im0 = (rb0 + j / nct) * TM4;
            in0 = (j % nct) * TN4;
        };
        open(0);
        auto issue = [&]() {                                       // slice ig of the stream -> stage ig % NST6; then advance
            unsigned char* sb = smem + (size_t)(ig % NST6) * STG6;
            const bool live = ig < S;                              // past the end: the last slice again (never read)
#pragma unroll
            for (int i = 0; i < PPP6; ++i) {
                const int p = q + NPROD * i;                       // (wave-uniform)
                const bool is_a = p < A6_PIECES;
                const int pl = is_a ? p : p - A6_PIECES;           // piece of its operand
                const int chunk = slot ^ (4 * (pl & 1) + (row_in >> 1));
                int row = (is_a ? im0 : in0) + pl * 8 + row_in;
                const int lim = is_a ? M : N;
                row = row < lim ? row : lim - 1;                   // rows beyond the matrix: any valid row (never stored)
                const unsigned short* src = (is_a ? A : Bt) + (size_t)row * Kp + (size_t)(live ? ikt : nk - 1) * BK6 + chunk * 8;
                unsigned char* dst = sb + (is_a ? 0 : A6_BYTES) + (size_t)pl * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16, 0, 0);
            }
            ++ig;
            if (live && ++ikt == nk && ig < S) {
                ikt = 0;
                open(++ii);
            }
        };
        // bare s_waitcnt + s_barrier: __syncthreads() carries vmcnt(0) and would wait for the slices just requested
#pragma unroll
        for (int g = 0; g < NST6 - 1; ++g) issue();
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NST6 - 2) * PPP6) : "memory");     // slice 0 landed
        issue();
        for (int g = 0; g < S; ++g) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NST6 - 2) * PPP6) : "memory"); // slice g + 1 landed; the stage of slice g is free
            issue();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // nothing may land after the workgroup's LDS is gone
        return;
    }

    // ---- compute wave: tile rows wr * 80 .., columns wc * 64 .. (5 x 4 MFMA tiles), two k-steps per slice
    const int wr = wave >> 2, wc = wave & 3;
    f32x4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment addresses: lane (r = lane & 15, g = lane >> 4) reads chunk (4 s + g) ^ ((r >> 1) & 7) of row (tile * 16 + r), 128-B rows;
    // tile i of a wave lies i * 2 KiB further on (immediate offset); the two k-steps of a slice differ in bit 2 of the slot: an XOR of 64 B
    const int fr = lane & 15, fg = lane >> 4;
    const unsigned lds0 = mg_lds_addr(smem);
    const unsigned aoff0 = lds0 + (unsigned)(((wr * 80 + fr) * 8 + (fg ^ ((fr >> 1) & 7))) * 16);
    const int bdelta = A6_BYTES + (wc * 64 - wr * 80) * 128;        // Bt fragment row of this lane - its A fragment row (wave-uniform)
    u32x4 a[2][5], b[2][4];
    auto reads = [&](int stage, int s2, int buf) {
        const unsigned ao = (aoff0 ^ (unsigned)(s2 * 64)) + (unsigned)stage * STG6, bo = ao + (unsigned)bdelta;
        a[buf][0] = mg_lds_read128<0>(ao);
        a[buf][1] = mg_lds_read128<2048>(ao);
        a[buf][2] = mg_lds_read128<4096>(ao);
        a[buf][3] = mg_lds_read128<6144>(ao);
        a[buf][4] = mg_lds_read128<8192>(ao);
        b[buf][0] = mg_lds_read128<0>(bo);
        b[buf][1] = mg_lds_read128<2048>(bo);
        b[buf][2] = mg_lds_read128<4096>(bo);
        b[buf][3] = mg_lds_read128<6144>(bo);
    };
    // tiles are computed transposed (A operand = Bt fragment) so a lane ends up with four consecutive C columns
    auto mmas = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[buf][jj]),
                                                                    __builtin_bit_cast(bf16x8, a[buf][i]), acc[i][jj], 0, 0, 0);
    };
    int ci = 0, ckt = 0;
    asm volatile("s_barrier" ::: "memory");                        // slice 0 landed (the producers waited for it)
    reads(0, 0, 0);
    // software pipeline (the 256 x 128 kernel's): the reads of k-step 1 go out before the MFMAs of k-step 0 and -- behind the slice barrier in
    // the MIDDLE of the iteration -- the reads of the next slice's k-step 0 before the MFMAs of k-step 1
    for (int g = 0; g < S; ++g) {
        reads(g % NST6, 1, 1);
        mg_lds_wait<9>();                                          // k-step 0 landed (the 9 reads of k-step 1 are behind it)
        __builtin_amdgcn_sched_barrier(0);
        mmas(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // done with the stage of slice g; slice g + 1 landed
        if (g + 1 < S) reads((g + 1) % NST6, 0, 0);
        if (g + 1 < S) mg_lds_wait<9>(); else mg_lds_wait<0>();
        __builtin_amdgcn_sched_barrier(0);
        mmas(1);
        __builtin_amdgcn_sched_barrier(0);
        if (++ckt < nk) continue;
        ckt = 0;
        // ---- epilogue: acc[i][jj][r] = C[cm0 + wr*80 + 16 i + (lane & 15)][cn0 + wc*64 + 16 jj + 4 (lane >> 4) + r]
        const int j = jj0 + ci * W;
        const int cm0 = (rb0 + j / nct) * TM4, cn0 = (j % nct) * TN4;
        ++ci;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int n = cn0 + wc * 64 + jj * 16 + fg * 4;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (bias && n < N) bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int m = cm0 + wr * 80 + i * 16 + fr;
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = mg_act(acc[i][jj][r] + bv[r], act);
                acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
                if ((c_bf16 & 2) && m < M && m < cm0 + TM4) {
                    // transposed store (round 6): C^T [N, ldc >= M] -- the 16 lanes of a row group hold 16 consecutive m of one n: 32 / 64 B
                    // runs per column.  A product with a SMALL M (the workload's W^T . X^T products, its read-out) runs as its transpose
                    // (M' = the long side: full rounds of tiles) and still leaves the K-contiguous operand the next product needs.
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (n + r >= N) continue;
                        if (c_bf16 & 1) {
                            unsigned pk;
                            asm("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(pk) : "v"(o[r]));
                            reinterpret_cast<unsigned short*>(C)[(size_t)(n + r) * ldc + m] = (unsigned short)(pk & 0xFFFFu);
                        } else {
                            C[(size_t)(n + r) * ldc + m] = o[r];
                        }
                    }
                } else if (!(c_bf16 & 2) && m < M && m < cm0 + TM4 && n < N) {
                    if (c_bf16 & 1) {
                        unsigned lo, hi;
                        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(o[0]), "v"(o[1]));
                        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(o[2]), "v"(o[3]));
                        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<u32x2_t*>(reinterpret_cast<unsigned short*>(C) + (size_t)m * ldc + n) = u32x2_t{lo, hi};
                    } else {
                        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(C + (size_t)m * ldc + n));
                    }
                }
            }
        }
    }
}

}  // namespace

// ---- 320 x 256 tiles, sixteen waves (round 5) -------------------------------------------------------------------------------
// The F = 2048 adjacency product (10 000 x 2048 x 10 000) on 256 x 256 tiles is 320 tiles = 1.25 rounds (left-over tiles cut along K,
// a fix-up launch) and 3.28 GB of L2 -> LDS traffic.  320 x 256 tiles: 32 row blocks x 8 column tiles = 256 tiles = ONE round, 2.95 GB.
// 80 x 64 wave tiles (80 accumulators) x SIXTEEN waves = four per SIMD at <= 128 registers: no producer waves, but a wave issues
// only 3 requests per 32-wide slice (waves 0..11; 36 pieces of 16 rows x 64 B) between its 20 MFMAs and three other waves of its
// SIMD cover its waits -- the eight-wave 256 x 256 kernel loses ~850 of 1850 cycles per slice to the issue of its own requests.
// Fragments are single buffered (read, wait, multiply: the other waves of the SIMD fill the pipe meanwhile).  Four-stage ring of
// 36-KB slices; whole tiles round robin over the XCD's workgroups; operand rows in LDS as in the kernels above.
namespace {
constexpr int TM5 = 320, TN5 = 256, BK5 = 32, NST5 = 4;
constexpr int A5_BYTES = TM5 * BK5 * 2, B5_BYTES = TN5 * BK5 * 2, STG5 = A5_BYTES + B5_BYTES;          // 20 KB + 16 KB
constexpr int A5_PIECES = A5_BYTES / 1024, B5_PIECES = B5_BYTES / 1024;                                  // 20 + 16 = 36 = 12 waves x 3
constexpr int REQW5 = 12, RPW5 = (A5_PIECES + B5_PIECES) / REQW5;
static_assert(REQW5 * RPW5 == A5_PIECES + B5_PIECES, "every requesting wave issues the same number of pieces");
constexpr size_t SMEM5_BYTES = (size_t)NST5 * STG5;
constexpr int NTHR5 = 1024;

__global__ __launch_bounds__(NTHR5) void gemm_bf16_nt_320_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ Bt,
                                                                 int M, int N, int Kp, const float* __restrict__ bias,
                                                                 float* __restrict__ C, int ldc, int act, int nrb, int nct, int c_bf16) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, jj0 = blockIdx.x >> 3, W = gridDim.x >> 3;
    const int nk = Kp / BK5;
    const int rb0 = xcd * nrb / 8, T = ((xcd + 1) * nrb / 8 - rb0) * nct;
    const int nitem = jj0 < T ? (T - jj0 + W - 1) / W : 0;
    if (nitem == 0) return;
    const int S = nitem * nk;
    // ---- requests (waves 0..11): piece p = wave + 12 i of a slice: p < 20 = A piece p, else Bt piece p - 20; lane (row_in = lane >> 2,
    //      slot = lane & 3) fetches the chunk that belongs in its slot; rows beyond M / N read zeros (buffer bounds)
    const bool requester = wave < REQW5;
    const int row_in = lane >> 2, slot = lane & 3;
    const int chunk = slot ^ ((0xD2 >> (2 * ((row_in >> 2) & 3))) & 3);
    const unsigned voff = (unsigned)((row_in * Kp + chunk * 8) * 2);
    int ii = 0, ik = 0;
    __amdgpu_buffer_rsrc_t ra, rb;
    auto open_tile = [&](int i) {
        const int j = jj0 + i * W;
        const int m0 = (rb0 + j / nct) * TM5, n0 = (j % nct) * TN5;
        const int mr = M - m0 < TM5 ? M - m0 : TM5, nr = N - n0 < TN5 ? N - n0 : TN5;
        ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A + (size_t)m0 * Kp), 0, mr * Kp * 2, 0x00027000);
        rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Bt + (size_t)n0 * Kp), 0, nr * Kp * 2, 0x00027000);
    };
    open_tile(0);
    auto issue_one = [&](int g, int i) {                           // request i (0..2) of slice g, k slice ik of the open tile
        unsigned char* sb = smem + (size_t)(g & (NST5 - 1)) * STG5;
        const int p = wave + REQW5 * i;                            // (wave-uniform)
        if (p < A5_PIECES) {
            const unsigned soff = (unsigned)((p * 16 * Kp + ik * BK5) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(uintptr_t)(sb + (size_t)p * 1024), 16, voff, soff, 0, 0);
        } else {
            const unsigned soff = (unsigned)(((p - A5_PIECES) * 16 * Kp + ik * BK5) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(uintptr_t)(sb + A5_BYTES + (size_t)(p - A5_PIECES) * 1024), 16, voff, soff, 0, 0);
        }
    };
    auto issue_done = [&]() {
        if (++ik == nk && ++ii < nitem) {
            ik = 0;
            open_tile(ii);
        }
    };
    // ---- compute: wave tile rows wr * 80 .., columns wc * 64 .. (5 x 4 MFMA tiles)
    const int wr = wave >> 2, wc = wave & 3;
    f32x4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    const unsigned lds0 = mg_lds_addr(smem);
    const unsigned swz = (unsigned)((fg ^ ((0xD2 >> (2 * ((fr >> 2) & 3))) & 3)) << 4);
    const unsigned aoff = lds0 + (unsigned)((wr * 80 + fr) * 64) + swz, boff = lds0 + A5_BYTES + (unsigned)((wc * 64 + fr) * 64) + swz;
    u32x4 a[5], b[4];
    if (requester) {
        for (int g = 0; g < NST5 - 1; ++g) {                       // (S >= 10: the launcher asks for it)
#pragma unroll
            for (int i = 0; i < RPW5; ++i) issue_one(g, i);
            issue_done();
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST5 - 2) * RPW5) : "memory");      // this wave's pieces of slice 0 landed
    }
    asm volatile("s_barrier" ::: "memory");                        // everybody's
    int ci = 0, ck = 0;
    for (int g = 0; g < S; ++g) {
        const bool req = requester && g + NST5 - 1 < S;            // slice g + 3 -> the stage of slice g - 1: read by everybody before the last barrier
        const unsigned so = (unsigned)(g & (NST5 - 1)) * STG5;
        b[0] = mg_lds_read128<0>(boff + so);
        b[1] = mg_lds_read128<1024>(boff + so);
        b[2] = mg_lds_read128<2048>(boff + so);
        b[3] = mg_lds_read128<3072>(boff + so);
        a[0] = mg_lds_read128<0>(aoff + so);
        a[1] = mg_lds_read128<1024>(aoff + so);
        a[2] = mg_lds_read128<2048>(aoff + so);
        a[3] = mg_lds_read128<3072>(aoff + so);
        a[4] = mg_lds_read128<4096>(aoff + so);
        mg_lds_wait<0>();
        __builtin_amdgcn_sched_barrier(0);
        // tiles are computed transposed (A operand = Bt fragment) so a lane ends up with four consecutive C columns; the wave's three
        // requests of slice g + 3 ride behind the MFMAs of rows 0, 2 and 4
#pragma unroll
        for (int i = 0; i < 5; ++i) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[jj]), __builtin_bit_cast(bf16x8, a[i]),
                                                                    acc[i][jj], 0, 0, 0);
            if (req && (i & 1) == 0) issue_one(g + NST5 - 1, i >> 1);
        }
        if (req) issue_done();
        __builtin_amdgcn_sched_barrier(0);
        // slice g + 1 landed (this wave's pieces: at most those of slices g + 2, g + 3 are younger); everybody is done with slice g
        if (requester) {
            if (req) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * RPW5) : "memory");
            else if (g + 2 < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RPW5) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");
        if (++ck < nk) continue;
        ck = 0;
        // ---- epilogue: acc[i][jj][r] = C[cm0 + wr*80 + 16 i + (lane & 15)][cn0 + wc*64 + 16 jj + 4 (lane >> 4) + r]
        const int j = jj0 + ci * W;
        const int cm0 = (rb0 + j / nct) * TM5, cn0 = (j % nct) * TN5;
        ++ci;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int n = cn0 + wc * 64 + jj * 16 + fg * 4;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (bias && n < N) bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int m = cm0 + wr * 80 + i * 16 + fr;
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = mg_act(acc[i][jj][r] + bv[r], act);
                acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (m < M && n < N) {
                    if (c_bf16) {                                  // C is a bf16 matrix (ldc in elements): the operand of the next product
                        unsigned lo, hi;
                        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(o[0]), "v"(o[1]));
                        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(o[2]), "v"(o[3]));
                        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<u32x2_t*>(reinterpret_cast<unsigned short*>(C) + (size_t)m * ldc + n) = u32x2_t{lo, hi};
                    } else {
                        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(C + (size_t)m * ldc + n));
                    }
                }
            }
        }
    }
}

}  // namespace

// ---- 320 x 256 tiles, 64-wide K slices (round 6) ---------------------------------------------------------------------------------
// The same step as gemm_bf16_nt_160k_kernel for the sixteen-wave tile: whole 128-B operand lines (half the L2 requests per byte).  A slice
// is 72 KB here, so the ring has TWO stages: the requests of slice g + 1 go out at the head of slice g (its stage was read for the last time
// before the barrier that closed slice g - 1) and must have landed at its end -- one slice in flight instead of three; what that costs in
// overlap the halved request count has to win back (measured: NOTES_r06 2a').  Pieces of 8 rows x 128 B, 72 per slice = 12 requesting
// waves x 6; operand rows in LDS as in the 256 x 128 kernel; fragments single buffered, two k-steps per slice.
namespace {
constexpr int BK7 = 64, NST7 = 2;
constexpr int A7_BYTES = TM5 * BK7 * 2, B7_BYTES = TN5 * BK7 * 2, STG7 = A7_BYTES + B7_BYTES;          // 40 KB + 32 KB
constexpr int A7_PIECES = A7_BYTES / 1024, B7_PIECES = B7_BYTES / 1024;                                  // 40 + 32 = 72
constexpr int RPW7 = (A7_PIECES + B7_PIECES) / REQW5;
static_assert(REQW5 * RPW7 == A7_PIECES + B7_PIECES && REQW5 % 2 == 0, "every requesting wave issues the same number of pieces, all of its own parity");
constexpr size_t SMEM7_BYTES = (size_t)NST7 * STG7;
static_assert(SMEM7_BYTES <= 160 * 1024, "LDS");

__global__ __launch_bounds__(NTHR5) void gemm_bf16_nt_320k_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ Bt,
                                                                  int M, int N, int Kp, const float* __restrict__ bias,
                                                                  float* __restrict__ C, int ldc, int act, int nrb, int nct, int c_bf16) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];      // (the k-step XOR below relies on 128-B aligned rows)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, jj0 = blockIdx.x >> 3, W = gridDim.x >> 3;
    const int nk = Kp / BK7;
    const int rb0 = xcd * nrb / 8, T = ((xcd + 1) * nrb / 8 - rb0) * nct;
    const int nitem = jj0 < T ? (T - jj0 + W - 1) / W : 0;
    if (nitem == 0) return;
    const int S = nitem * nk;
    // ---- requests (waves 0..11): piece p = wave + 12 i of a slice: p < 40 = A piece p, else Bt piece p - 40 (same parity as the wave);
    //      lane (row_in = lane >> 3, slot = lane & 7) fetches the chunk that belongs in its slot: slot = chunk ^ ((row >> 1) & 7) with
    //      (row >> 1) & 7 = 4 (piece & 1) + (row_in >> 1); rows beyond M / N read zeros (buffer bounds)
    const bool requester = wave < REQW5;
    const int row_in = lane >> 3, slot = lane & 7;
    const int chunk = slot ^ (4 * (wave & 1) + (row_in >> 1));
    const unsigned voff = (unsigned)((row_in * Kp + chunk * 8) * 2);
    int ii = 0, ik = 0;
    __amdgpu_buffer_rsrc_t ra, rb;
    auto open_tile = [&](int i) {
        const int j = jj0 + i * W;
        const int m0 = (rb0 + j / nct) * TM5, n0 = (j % nct) * TN5;
        const int mr = M - m0 < TM5 ? M - m0 : TM5, nr = N - n0 < TN5 ? N - n0 : TN5;
        ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A + (size_t)m0 * Kp), 0, mr * Kp * 2, 0x00027000);
        rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Bt + (size_t)n0 * Kp), 0, nr * Kp * 2, 0x00027000);
    };
    open_tile(0);
    auto issue_one = [&](int g, int i) {                           // request i (0..5) of slice g, k slice ik of the open tile
        unsigned char* sb = smem + (size_t)(g & 1) * STG7;
        const int p = wave + REQW5 * i;                            // (wave-uniform)
        if (p < A7_PIECES) {
            const unsigned soff = (unsigned)((p * 8 * Kp + ik * BK7) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(uintptr_t)(sb + (size_t)p * 1024), 16, voff, soff, 0, 0);
        } else {
            const unsigned soff = (unsigned)(((p - A7_PIECES) * 8 * Kp + ik * BK7) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(uintptr_t)(sb + A7_BYTES + (size_t)(p - A7_PIECES) * 1024), 16, voff, soff, 0, 0);
        }
    };
    auto issue_done = [&]() {
        if (++ik == nk && ++ii < nitem) {
            ik = 0;
            open_tile(ii);
        }
    };
    // ---- compute: wave tile rows wr * 80 .., columns wc * 64 .. (5 x 4 MFMA tiles)
    const int wr = wave >> 2, wc = wave & 3;
    f32x4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    const unsigned lds0 = mg_lds_addr(smem);
    const unsigned aoff0 = lds0 + (unsigned)(((wr * 80 + fr) * 8 + (fg ^ ((fr >> 1) & 7))) * 16);
    const int bdelta = A7_BYTES + (wc * 64 - wr * 80) * 128;        // Bt fragment row of this lane - its A fragment row (wave-uniform)
    u32x4 a[5], b[4];
    if (requester) {
#pragma unroll
        for (int i = 0; i < RPW7; ++i) issue_one(0, i);
        issue_done();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces of slice 0 landed
    }
    asm volatile("s_barrier" ::: "memory");                        // everybody's
    int ci = 0, ck = 0;
    for (int g = 0; g < S; ++g) {
        const bool req = requester && g + 1 < S;                   // slice g + 1 -> the stage of slice g - 1: read by everybody before the last barrier
        const unsigned so = (unsigned)(g & 1) * STG7;
        mg_mha_static_for_2([&](auto sc) {
            constexpr int s2 = decltype(sc)::v;
            const unsigned ao = (aoff0 ^ (unsigned)(s2 * 64)) + so, bo = ao + (unsigned)bdelta;
            b[0] = mg_lds_read128<0>(bo);
            b[1] = mg_lds_read128<2048>(bo);
            b[2] = mg_lds_read128<4096>(bo);
            b[3] = mg_lds_read128<6144>(bo);
            a[0] = mg_lds_read128<0>(ao);
            a[1] = mg_lds_read128<2048>(ao);
            a[2] = mg_lds_read128<4096>(ao);
            a[3] = mg_lds_read128<6144>(ao);
            a[4] = mg_lds_read128<8192>(ao);
            mg_lds_wait<0>();
            __builtin_amdgcn_sched_barrier(0);
            // tiles are computed transposed (A operand = Bt fragment) so a lane ends up with four consecutive C columns; the wave's six
            // requests of slice g + 1 ride behind the MFMAs of rows 0, 2 and 4 of both k-steps
#pragma unroll
            for (int i = 0; i < 5; ++i) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[jj]), __builtin_bit_cast(bf16x8, a[i]),
                                                                        acc[i][jj], 0, 0, 0);
                if (req && (i & 1) == 0) issue_one(g + 1, 3 * s2 + (i >> 1));
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (req) issue_done();
        // slice g + 1 landed (this wave's pieces); everybody is done with slice g
        if (requester) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (++ck < nk) continue;
        ck = 0;
        // ---- epilogue: acc[i][jj][r] = C[cm0 + wr*80 + 16 i + (lane & 15)][cn0 + wc*64 + 16 jj + 4 (lane >> 4) + r]
        const int j = jj0 + ci * W;
        const int cm0 = (rb0 + j / nct) * TM5, cn0 = (j % nct) * TN5;
        ++ci;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int n = cn0 + wc * 64 + jj * 16 + fg * 4;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (bias && n < N) bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int m = cm0 + wr * 80 + i * 16 + fr;
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = mg_act(acc[i][jj][r] + bv[r], act);
                acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (m < M && n < N) {
                    if (c_bf16) {
                        unsigned lo, hi;
                        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(o[0]), "v"(o[1]));
                        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(o[2]), "v"(o[3]));
                        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<u32x2_t*>(reinterpret_cast<unsigned short*>(C) + (size_t)m * ldc + n) = u32x2_t{lo, hi};
                    } else {
                        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(C + (size_t)m * ldc + n));
                    }
                }
            }
        }
    }
}

}  // namespace

// Which round-5 tile shape a product takes: 4 = 160 x 256, 5 = 320 x 256, 0 = neither (round 4's kernels decide between themselves
// below).  Pure host arithmetic (no device call): mgnns_gemm_bf16_pick_form exposes it to tests.
//   want: 1 / 3 = the 160 x 256 / 320 x 256 kernel whenever the shape fits it, 2 = by the estimate, 0 = neither.
static int mg_gemm_pick(int M, int N, int Kp, bool ws_ok, size_t workspace_bytes, int n_cu, int want, bool allow256) {
    if (want == 0 || n_cu <= 0) return 0;
    const int per4 = n_cu / 8 > 0 ? n_cu / 8 : 1;
    const int nrb4 = (M + TM4 - 1) / TM4, nct4 = (N + TN4 - 1) / TN4, nrb5 = (M + TM5 - 1) / TM5, nct5 = (N + TN5 - 1) / TN5;
    const bool fits4 = nrb4 >= 8 && N >= TN4 && Kp / BK4 >= 2 * NST4, fits5 = nrb5 >= 8 && N >= TN5 && Kp / BK5 >= 10;
    int pick = want == 1 && fits4 ? 4 : (want == 3 && fits5 ? 5 : 0);
    if (want != 2 || !(fits4 || fits5)) return pick;
    // cost of a form = rounds of the busiest XCD x (32-wide slices x operand rows per tile-slice + the tile's fixed cost in the same
    // unit); ~1.7 ns per row-slice for every kernel (they stream at what the L2s deliver), fixed costs fitted at K = 320 (3.6 / 8.8 /
    // 15 us per tile: pipeline fill + an epilogue that nothing overlaps).  A last round that is at most a quarter full is cut along K
    // by round 4's kernels (+ ~1/16 round for the fix-up launch) if a part keeps >= 8 slices.
    const int nk64 = Kp / BK;
    auto rounds = [&](int t, bool ksplit) {
        const int full = t / per4, rem = t % per4;
        if (rem == 0) return (double)full;
        const int f = per4 / rem > 8 ? 8 : per4 / rem;
        return full + (ksplit && 4 * rem <= per4 && f * 8 <= nk64 ? 1.0 / f + 0.0625 : 1.0);
    };
    auto cost = [&](double r, int rows, int fixed) { return r * ((double)(Kp / BK5) * rows + fixed); };
    const int nrb1 = (M + TM - 1) / TM, nct1 = (N + TN - 1) / TN, nrb2 = (M + TM2 - 1) / TM2, nct2 = (N + TN2 - 1) / TN2;
    double best = cost(rounds(((nrb1 + 7) / 8) * nct1, ws_ok), TM + TN, 2100);
    const bool elig256 = ws_ok && allow256 && nrb2 / 8 * nct2 >= per4 && N >= TN2 && Kp / BK >= 64 &&
                         workspace_bytes >= (size_t)8 * per4 * TM2 * TN2 * sizeof(float);
    if (elig256) {
        const double c256 = cost(rounds(((nrb2 + 7) / 8) * nct2, true), TM2 + TN2, 8000);
        best = c256 < best ? c256 : best;
    }
    // measured, cache-cold, us (round 4's kernels / 160 x 256 / 320 x 256): 10 000 x 1024 x 10 000: 229-264 / 215-226 / 277;
    // 10 000 x 2048 x 10 000: 380-412 / 426-430 / 320; 10 000 x 1024 x 2048: 89-92 / 56-57 / 66; 10 000 x 2048 x 1024: 65-70 /
    // 62-64 / 53; 10 000 x 1024 x 320: 25.6 / 22.0 / -; 20 154 x 1200 x 320 (the LSTM's folded table): 40 / 47 / 51
    if (fits4) {
        const double c4 = cost(rounds(((nrb4 + 7) / 8) * nct4, false), TM4 + TN4, 5200) * 1.02;
        if (c4 < best) {
            best = c4;
            pick = 4;
        }
    }
    if (fits5) {
        const double c5 = cost(rounds(((nrb5 + 7) / 8) * nct5, false), TM5 + TN5, 8800) * 1.02;
        if (c5 < best) {
            best = c5;
            pick = 5;
        }
    }
    return pick;
}

extern "C" int mgnns_gemm_bf16_pick_form(int M, int N, int Kp, int with_workspace, int n_cu) {
    MG_REQUIRE(M > 0 && N > 0 && Kp > 0 && Kp % BK == 0 && n_cu > 0, "mgnns_gemm_bf16_pick_form: M=%d N=%d Kp=%d n_cu=%d", M, N, Kp, n_cu);
    return mg_gemm_pick(M, N, Kp, with_workspace != 0, with_workspace ? (size_t)256 * TM2 * TN2 * sizeof(float) : 0, n_cu, 2, true);
}

static int g_gemm160_bk = 0;            // K-slice width of the 160 x 256 tile: 0 = MGNNS_GEMM160_BK (default 64), 32 / 64 forced (mgnns_gemm_bf16_set_form(101 / 102))
static int g_gemm_form = -1;            // -1: MGNNS_GEMM_160 (default 2); 0 round 4's kernels only, 1 / 3 the 160 x 256 / 320 x 256 kernel whenever the shape fits, 2 by the estimate
extern "C" int mgnns_gemm_bf16_set_form(int form) {
    if (form >= 100 && form <= 102) {                   // K-slice width of the 160 x 256 tile: 100 environment (MGNNS_GEMM160_BK, default 64), 101 = 32, 102 = 64
        g_gemm160_bk = form == 100 ? 0 : form == 101 ? 32 : 64;
        return 0;
    }
    MG_REQUIRE(form >= -1 && form <= 3, "mgnns_gemm_bf16_set_form: form=%d (-1 environment, 0 round 4's kernels only, 1 160 x 256 whenever it fits, 2 by estimate, 3 320 x 256 whenever it fits)", form);
    g_gemm_form = form;
    return 0;
}

// internal launcher (also used by the bf16-mode LSTM input projections): m_dev != nullptr -> the row count is read on the
// device and M is only its upper bound
int mg_launch_gemm_bf16(const void* A, const void* Bt, int M, int N, int Kp, const float* bias, float* C, int ldc, int act,
                        const int32_t* m_dev, hipStream_t stream, int c_bf16, void* workspace, size_t workspace_bytes) {
    MG_REQUIRE(A && Bt && C, "mgnns_gemm_bf16_nt_fwd: null pointer");
    const bool c_tr = (c_bf16 & 2) != 0;                           // C^T [N, ldc >= M] instead of C [M, ldc >= N]
    MG_REQUIRE(M >= 0 && N > 0 && N % 4 == 0 && Kp > 0 && Kp % BK == 0 && ldc >= (c_tr ? M : N) && (c_tr || ldc % 4 == 0),
               "mgnns_gemm_bf16_nt_fwd: need N %% 4 == 0, Kp %% %d == 0, ldc %% 4 == 0 (M=%d N=%d Kp=%d ldc=%d)", BK, M, N, Kp, ldc);
    MG_REQUIRE(!c_tr || (!m_dev && (M + TM4 - 1) / TM4 >= 8 && N >= TN4 && Kp / BK4 >= 2 * NST4),
               "mgnns_gemm_bf16_nt_fwd: the transposed store is the 160 x 256 kernel's (M=%d >= 1121, N=%d >= 256, K=%d >= 320)", M, N, Kp);
    MG_REQUIRE(act >= 0 && act <= 2, "mgnns_gemm_bf16_nt_fwd: unknown activation %d", act);
    MG_REQUIRE(mg_aligned16(A) && mg_aligned16(Bt) && mg_aligned16(C) && (!bias || mg_aligned16(bias)),
               "mgnns_gemm_bf16_nt_fwd: operands must be 16-byte aligned");
    if (M == 0) return 0;
    // Round 5's tile shapes -- 160 x 256 with producer waves, 320 x 256 with sixteen waves -- where they fill the chip's rounds
    // better than the other kernels' tiles do: the estimate is rounds of the busiest XCD x operand rows per tile-slice (all of them
    // stream at about what the L2s deliver).  MGNNS_GEMM_160 / mgnns_gemm_bf16_set_form: 0 neither, 1 160 x 256 whenever the shape
    // fits, 2 (default) by the estimate, 3 320 x 256 whenever the shape fits
    if (const int want = m_dev ? 0 : c_tr ? 1 : (g_gemm_form >= 0 ? g_gemm_form : mg_env_int("MGNNS_GEMM_160", 2, 9))) {
        const int n_cu4 = mg_cu_count();
        if (n_cu4 <= 0) return MGNNS_ERR_LAUNCH;
        const int per4 = n_cu4 / 8 > 0 ? n_cu4 / 8 : 1;
        const int nrb4 = (M + TM4 - 1) / TM4, nct4 = (N + TN4 - 1) / TN4, nrb5 = (M + TM5 - 1) / TM5, nct5 = (N + TN5 - 1) / TN5;
        const int pick = c_tr ? 4 : mg_gemm_pick(M, N, Kp, workspace && mg_aligned16(workspace), workspace_bytes, n_cu4, want,
                                                 mg_env_int("MGNNS_GEMM_TILE", 256, 2) == 256);
        if (pick == 4 && Kp % BK6 == 0 && (c_tr || (g_gemm160_bk > 0 ? g_gemm160_bk : mg_env_int("MGNNS_GEMM160_BK", 64, 11)) == 64)) {
            MG_DYN_LDS(gemm_bf16_nt_160k_kernel, SMEM6_BYTES);
            hipLaunchKernelGGL(gemm_bf16_nt_160k_kernel, dim3(8 * per4), dim3(NTHR4), SMEM6_BYTES, stream,
                               reinterpret_cast<const unsigned short*>(A), reinterpret_cast<const unsigned short*>(Bt), M, N, Kp, bias, C,
                               ldc, act, nrb4, nct4, c_bf16);
            MG_CHECK_LAUNCH("mgnns_gemm_bf16_nt_fwd(160, 64-wide slices)");
            return 0;
        }
        if (pick == 4) {
            MG_DYN_LDS(gemm_bf16_nt_160_kernel, SMEM4_BYTES);
            hipLaunchKernelGGL(gemm_bf16_nt_160_kernel, dim3(8 * per4), dim3(NTHR4), SMEM4_BYTES, stream,
                               reinterpret_cast<const unsigned short*>(A), reinterpret_cast<const unsigned short*>(Bt), M, N, Kp, bias, C,
                               ldc, act, nrb4, nct4, c_bf16);
            MG_CHECK_LAUNCH("mgnns_gemm_bf16_nt_fwd(160)");
            return 0;
        }
        if (pick == 5 && Kp % BK7 == 0 && (g_gemm160_bk > 0 ? g_gemm160_bk : mg_env_int("MGNNS_GEMM160_BK", 64, 11)) == 64 &&
            mg_env_int("MGNNS_GEMM320_BK", 64, 12) == 64) {
            MG_DYN_LDS(gemm_bf16_nt_320k_kernel, SMEM7_BYTES);
            hipLaunchKernelGGL(gemm_bf16_nt_320k_kernel, dim3(8 * per4), dim3(NTHR5), SMEM7_BYTES, stream,
                               reinterpret_cast<const unsigned short*>(A), reinterpret_cast<const unsigned short*>(Bt), M, N, Kp, bias, C,
                               ldc, act, nrb5, nct5, c_bf16);
            MG_CHECK_LAUNCH("mgnns_gemm_bf16_nt_fwd(320, 64-wide slices)");
            return 0;
        }
        if (pick == 5) {
            MG_DYN_LDS(gemm_bf16_nt_320_kernel, SMEM5_BYTES);
            hipLaunchKernelGGL(gemm_bf16_nt_320_kernel, dim3(8 * per4), dim3(NTHR5), SMEM5_BYTES, stream,
                               reinterpret_cast<const unsigned short*>(A), reinterpret_cast<const unsigned short*>(Bt), M, N, Kp, bias, C,
                               ldc, act, nrb5, nct5, c_bf16);
            MG_CHECK_LAUNCH("mgnns_gemm_bf16_nt_fwd(320)");
            return 0;
        }
    }
    // at least 128 tiles of 256 x 256 (half a chip of workgroups) and the workspace for its pieces: the form with a third less
    // L2 -> LDS traffic (MGNNS_GEMM_TILE=128: the 256 x 128 kernel)
    if (!m_dev && workspace && mg_aligned16(workspace) && mg_env_int("MGNNS_GEMM_TILE", 256, 2) == 256) {     // (=128: the 256 x 128 kernel only)
        const int nrb2 = (M + TM2 - 1) / TM2, nct2 = (N + TN2 - 1) / TN2;
        const int n_cu2 = mg_cu_count();
        if (n_cu2 <= 0) return MGNNS_ERR_LAUNCH;
        int per = n_cu2 / 8;
        if (per < 1) per = 1;
        // at least one full round of tiles on every XCD (fewer: the 256 x 128 kernel fills the chip better) and a long K (measured:
        // 10 000 x 2048 with K = 1024 / 2048, the X.W products of configs[4], lose 20 % -- the fixed cost per tile counts there)
        if (nrb2 / 8 * nct2 >= per && N >= TN2 && Kp / BK >= 64 && workspace_bytes >= (size_t)8 * per * TM2 * TN2 * sizeof(float)) {
            MG_DYN_LDS(gemm_bf16_nt_256_kernel, SMEM3_BYTES);
            hipLaunchKernelGGL(gemm_bf16_nt_256_kernel, dim3(8 * per), dim3(512), SMEM3_BYTES, stream,
                               reinterpret_cast<const unsigned short*>(A), reinterpret_cast<const unsigned short*>(Bt), M, N, Kp, bias, C,
                               ldc, act, nrb2, nct2, c_bf16, static_cast<float*>(workspace));
            MG_CHECK_LAUNCH("mgnns_gemm_bf16_nt_fwd(256)");
            int max_rem = 0;
            for (int x = 0; x < 8; ++x) {
                const int Tx = ((x + 1) * nrb2 / 8 - x * nrb2 / 8) * nct2;
                int full, rem, f;
                gemm256_split(Tx, per, Kp / BK, full, rem, f);
                if (f >= 2 && rem > max_rem) max_rem = rem;
            }
            if (max_rem) {
                hipLaunchKernelGGL(gemm_bf16_fixup_256_kernel, dim3(8 * max_rem), dim3(512), 0, stream, static_cast<const float*>(workspace), M, N,
                                   Kp, bias, C, ldc, act, nrb2, nct2, per, c_bf16);
                MG_CHECK_LAUNCH("mgnns_gemm_bf16_nt_fwd(256 fix-up)");
            }
            return 0;
        }
    }
    MG_DYN_LDS(gemm_bf16_nt_kernel, SMEM_BYTES);
    const int nrb = (M + TM - 1) / TM, nct = (N + TN - 1) / TN;
    int rps = 1;                                                   // row-block ranges per XCD (see the kernel)
    int jmax = ((nrb + 7) / 8) * nct;                              // virtual tiles per XCD
    if (nrb < 8 && nct >= 8) {                                     // fewer row blocks than XCDs: split the column tiles over the XCDs
        rps = 0;
        jmax = nrb * ((nct + 7) / 8);
    }
    const int n_cu = mg_cu_count();
    if (n_cu <= 0) return MGNNS_ERR_LAUNCH;
    int per_xcd = n_cu / 8;                                        // one persistent workgroup per CU
    if (per_xcd < 1) per_xcd = 1;
    if (per_xcd > jmax) per_xcd = jmax;
    // workspace (optional): 8 W slots of 256 x 128 fp32 partial sums for the K split of a last, partial round of tiles (see the
    // kernel; mg_gemm_split decides here, too, whether there is anything to fix up)
    float* ws_part = nullptr;
    if (workspace && !m_dev && workspace_bytes >= (size_t)8 * per_xcd * TM * TN * sizeof(float) && mg_aligned16(workspace))
        ws_part = static_cast<float*>(workspace);
    const int W = per_xcd, nk = Kp / BK;
    int max_rem = 0;
    if (ws_part) {
        for (int xcd = 0; xcd < 8; ++xcd) {
            int T, full, rem, f;
            mg_gemm_split(xcd, nrb, nct, rps, jmax, W, nk, T, full, rem, f);
            if (f && rem > max_rem) max_rem = rem;
        }
        if (!max_rem) ws_part = nullptr;                           // no XCD has a splittable last round
    }
    hipLaunchKernelGGL(gemm_bf16_nt_kernel, dim3(8 * per_xcd), dim3(NTHR_WS), SMEM_BYTES, stream,
                       reinterpret_cast<const unsigned short*>(A), reinterpret_cast<const unsigned short*>(Bt), M, N, Kp, bias, C,
                       ldc, act, nrb, nct, rps, jmax, m_dev, c_bf16, ws_part);
    MG_CHECK_LAUNCH("mgnns_gemm_bf16_nt_fwd");
    if (ws_part) {
        hipLaunchKernelGGL(gemm_bf16_fixup_kernel, dim3(8 * max_rem), dim3(512), 0, stream, (const float*)ws_part, M, N, bias, C, ldc, act,
                           nrb, nct, rps, jmax, W, nk, c_bf16);
        MG_CHECK_LAUNCH("mgnns_gemm_bf16_nt_fwd(fix-up)");
    }
    return 0;
}

// 256 workgroups x one part of 256 x 256 fp32 (the 256 x 256 kernel); the 256 x 128 kernel's K split uses the first half
#ifdef MG_GEMM_TRACE
extern "C" int mgnns_debug_gemm_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_trace), sizeof(unsigned long long) * 4) == hipSuccess ? 0 : 1;
}
#endif

extern "C" size_t mgnns_gemm_bf16_workspace_bytes(void) { return (size_t)256 * TM2 * TN2 * sizeof(float); }

extern "C" int mgnns_gemm_bf16_nt_fwd(const void* A, const void* Bt, int M, int N, int Kp, const float* bias, void* C, int ldc,
                                      int c_bf16, int act, void* workspace, size_t workspace_bytes, mgnns_stream_t stream) {
    MG_REQUIRE(c_bf16 >= 0 && c_bf16 <= 3, "mgnns_gemm_bf16_nt_fwd: c_bf16=%d (bit 0: bf16 output, bit 1: transposed store)", c_bf16);
    return mg_launch_gemm_bf16(A, Bt, M, N, Kp, bias, static_cast<float*>(C), ldc, act, nullptr, (hipStream_t)stream, c_bf16,
                               workspace, workspace_bytes);
}
