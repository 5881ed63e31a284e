// Dense bf16 GEMM on v_mfma_f32_16x16x32_bf16, fp32 accumulation and output (BASELINE configs[4] (i): the dense
// [N,N] adjacency path of the label / vocabulary graph GCN, UTIL:421-426 `adj @ support`, and any large X.W):
//   C[M,N] = act(A[M,K] . Bt[N,K]^T + bias),  A and Bt bf16, K-contiguous rows of Kp elements (Kp % 64 == 0, zero padded)
//
// Workgroup tile 256 x 128, eight waves 4 x 2 = two per SIMD (wave tile 64 x 64 = 4 x 4 MFMA tiles; with one wave per
// SIMD and a 128 x 64 wave tile the MFMAs issued at ~36 cycles instead of ~17 next to the fragment reads),
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

constexpr int TM = 256, TN = 128, BK = 64, NSTAGE = 3, NTHR = 512, NWAVE = NTHR / 64;
constexpr int A_BYTES = TM * BK * 2, B_BYTES = TN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;   // 32 KB + 16 KB
constexpr int PIECES = STAGE_BYTES / 1024, PPW = PIECES / NWAVE;                               // 48 DMA pieces, 6 per wave
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

__global__ __launch_bounds__(NTHR) void gemm_bf16_nt_kernel(const unsigned short* __restrict__ A,
                                                            const unsigned short* __restrict__ Bt, int M, int N, int Kp,
                                                            const float* __restrict__ bias, float* __restrict__ C, int ldc,
                                                            int act, int nrb, int nct, int rps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- XCD-aware tile map
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int per = rps * nct;
    const int sl = j / per, within = j - sl * per;
    const int rb = (sl * 8 + xcd) * rps + within / nct, ct = within % nct;
    if (rb >= nrb) return;
    const int m0 = rb * TM, n0 = ct * TN;
    const int wr = wave >> 1, wc = wave & 1;                       // wave tile: rows wr*64.., cols wc*64..
    const int nk = Kp / BK;

    // ---- LDS-DMA of one BK slice: piece p covers 8 rows x 128 B; lane (row_in = lane >> 3, slot = lane & 7) fetches
    //      the chunk that belongs in its slot (rows r and r + 8 of a fragment sit 1 KiB apart = the same banks, so the
    //      swizzle has to tell them apart: it uses row bits 1..3; row bit 0 already selects the 128-B half of a bank row)
    const int row_in = lane >> 3, slot = lane & 7;
    auto issue = [&](int kt, int stage) {
        unsigned char* sb = smem + (size_t)stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + NWAVE * i;                       // wave-uniform
            // slot = chunk ^ ((row >> 1) & 7); a piece starts at a multiple of 8 rows: (row >> 1) & 7 = 4 (p & 1) + (row_in >> 1)
            const int chunk = slot ^ (4 * (p & 1) + (row_in >> 1));   // p & 1 == (p - 32) & 1: same rule for the Bt pieces
            const unsigned short* src;
            if (p < A_BYTES / 1024) {
                int row = m0 + p * 8 + row_in;
                row = row < M ? row : M - 1;                      // rows beyond M: any valid row (never stored)
                src = A + (size_t)row * Kp + (size_t)kt * BK + chunk * 8;
            } else {
                int row = n0 + (p - A_BYTES / 1024) * 8 + row_in;
                row = row < N ? row : N - 1;
                src = Bt + (size_t)row * Kp + (size_t)kt * BK + chunk * 8;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(uintptr_t)(sb + (size_t)p * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses: lane (r = lane & 15, g = lane >> 4) reads chunk (4 s + g) ^ ((r >> 1) & 7) of row (tile * 16 + r);
    // tile i of a wave lies i * 2 KiB further on (immediate offset).  The reads go through mg_lds_read128 (inline asm):
    // as ordinary ds_reads hipcc guards each group with s_waitcnt vmcnt(0) against the LDS-DMA in flight, i.e. it
    // waits for the slice that was requested a moment ago and the three-stage ring never overlaps anything.
    const int fr = lane & 15, fg = lane >> 4;
    const unsigned lds0 = mg_lds_addr(smem);
    unsigned aoff[2], boff[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        aoff[s2] = lds0 + ((wr * 64 + fr) * 8 + ((4 * s2 + fg) ^ ((fr >> 1) & 7))) * 16;
        boff[s2] = lds0 + A_BYTES + ((wc * 64 + fr) * 8 + ((4 * s2 + fg) ^ ((fr >> 1) & 7))) * 16;
    }

    // Software pipeline: fragments are double buffered per k-step; the reads of k-step 1 are issued before the MFMAs
    // of k-step 0, and -- behind the slice barrier placed in the MIDDLE of an iteration -- the reads of the NEXT
    // slice's k-step 0 before the MFMAs of k-step 1.  The barrier also frees the current slice's stage (both k-steps
    // are in registers by then) for the DMA of slice kt + 3.  Barriers are bare s_barrier + the s_waitcnt actually
    // needed: __syncthreads() carries a workgroup fence, i.e. s_waitcnt vmcnt(0), which would wait for DMA slices
    // that were only just requested.
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
    auto mmas = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[buf][jj]),
                                                                    __builtin_bit_cast(bf16x8, a[buf][i]), acc[i][jj], 0, 0, 0);
    };
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (nk > 2) issue(2, 2);
    reads(0, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        reads(kt % NSTAGE, 1, 1);
        mg_lds_wait<8>();                                     // k-step 0 landed (the 8 reads of k-step 1 are behind it)
        __builtin_amdgcn_sched_barrier(0);
        mmas(0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            // slice kt+1 landed: of this wave's DMA pieces at most those of slice kt+2 may still be in flight; this wave's
            // fragment reads of slice kt are complete (lgkmcnt) so its stage may be overwritten after the barrier
            if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // the two waves of a SIMD issue their DMA pieces (~100+ cycles each, MFMA issue blocked meanwhile) at
            // DIFFERENT points of the iteration: waves 0-3 here, waves 4-7 after their second k-step
            if (wave < 4 && kt + 3 < nk) issue(kt + 3, kt % NSTAGE);
            reads((kt + 1) % NSTAGE, 0, 0);
            mg_lds_wait<8>();                                 // k-step 1 landed at the barrier (lgkmcnt(0)); orders the MFMAs
        } else {
            mg_lds_wait<0>();
        }
        __builtin_amdgcn_sched_barrier(0);
        mmas(1);
        __builtin_amdgcn_sched_barrier(0);
        if (wave >= 4 && kt + 1 < nk && kt + 3 < nk) issue(kt + 3, kt % NSTAGE);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: acc[i][jj][r] = C[m0 + wr*64 + 16 i + (lane & 15)][n0 + wc*64 + 16 jj + 4 (lane >> 4) + r]
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int n = n0 + wc * 64 + jj * 16 + fg * 4;
        if (n >= N) continue;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias) bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wr * 64 + i * 16 + fr;
            if (m >= M) continue;
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = mg_act(acc[i][jj][r] + bv[r], act);
            *reinterpret_cast<f32x4*>(C + (size_t)m * ldc + n) = o;
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

extern "C" int mgnns_gemm_bf16_nt_fwd(const void* A, const void* Bt, int M, int N, int Kp, const float* bias, float* C, int ldc,
                                      int act, mgnns_stream_t stream) {
    MG_REQUIRE(A && Bt && C, "mgnns_gemm_bf16_nt_fwd: null pointer");
    MG_REQUIRE(M >= 0 && N > 0 && N % 4 == 0 && Kp > 0 && Kp % BK == 0 && ldc >= N && ldc % 4 == 0,
               "mgnns_gemm_bf16_nt_fwd: need N %% 4 == 0, Kp %% %d == 0, ldc %% 4 == 0 (M=%d N=%d Kp=%d ldc=%d)", BK, M, N, Kp, ldc);
    MG_REQUIRE(act >= 0 && act <= 2, "mgnns_gemm_bf16_nt_fwd: unknown activation %d", act);
    MG_REQUIRE(mg_aligned16(A) && mg_aligned16(Bt) && mg_aligned16(C) && (!bias || mg_aligned16(bias)),
               "mgnns_gemm_bf16_nt_fwd: operands must be 16-byte aligned");
    if (M == 0) return 0;
    MG_DYN_LDS(gemm_bf16_nt_kernel, SMEM_BYTES);
    const int nrb = (M + TM - 1) / TM, nct = (N + TN - 1) / TN;
    int rps = 32 / nct;
    if (rps < 1) rps = 1;
    if (rps > nrb) rps = nrb;
    const int supers = (nrb + rps - 1) / rps;
    const int blocks = 8 * ((supers + 7) / 8) * rps * nct;
    hipLaunchKernelGGL(gemm_bf16_nt_kernel, dim3(blocks), dim3(NTHR), SMEM_BYTES, (hipStream_t)stream,
                       reinterpret_cast<const unsigned short*>(A), reinterpret_cast<const unsigned short*>(Bt), M, N, Kp, bias, C,
                       ldc, act, nrb, nct, rps);
    MG_CHECK_LAUNCH("mgnns_gemm_bf16_nt_fwd");
    return 0;
}
