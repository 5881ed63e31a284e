// Adjacency normalisation (gen_adj, utils/util.py:421-426), dense -> CSR, and the CSR SpMM that is
// the "adj @ support" half of GraphConvolution.forward (Multi_GCN_Multihead_att.py:54).
#include "common.hpp"

namespace {

// d[i] = (sum_j A[i,j])^-1/2, one wave per row, coalesced row reads + wavefront shuffle reduction
__global__ __launch_bounds__(256) void rowsum_rsqrt_kernel(const float* __restrict__ A, int C,
                                                           float* __restrict__ d) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= C) return;
    const float* row = A + (size_t)i * C;
    float s = 0.f;
    for (int j = lane; j < C; j += 64) s += row[j];
    s = wave_sum(s);
    if (lane == 0) d[i] = powf(s, -0.5f);
}

// adj[i,j] = (A[j,i] * d[i]) * d[j]   -- the rounding order of ((A D)^T D)
__global__ __launch_bounds__(256) void gen_adj_kernel(const float* __restrict__ A, int C,
                                                      const float* __restrict__ d, float* __restrict__ adj) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;   // output tile rows i, cols j
    for (int r = ty; r < 32; r += 8) {                       // read A[j0+r, i0+tx]
        const int j = j0 + r, i = i0 + tx;
        tile[r][tx] = (j < C && i < C) ? A[(size_t)j * C + i] * d[i] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {                       // write adj[i0+r, j0+tx]
        const int i = i0 + r, j = j0 + tx;
        if (i < C && j < C) adj[(size_t)i * C + j] = tile[tx][r] * d[j];
    }
}

// nnz per row of a dense matrix; one wave per row
__global__ __launch_bounds__(256) void row_nnz_kernel(const float* __restrict__ M, int C, int32_t* __restrict__ cnt) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= C) return;
    const float* row = M + (size_t)i * C;
    int n = 0;
    for (int j0 = 0; j0 < C; j0 += 64) {
        const int j = j0 + lane;
        const bool nz = j < C && row[j] != 0.0f;
        n += __popcll(__ballot(nz));
    }
    if (lane == 0) cnt[i + 1] = n;
}

// in-place inclusive scan of cnt[1..C] (cnt[0] = 0): single block, C-long serial chunks per thread
__global__ __launch_bounds__(1024) void scan_kernel(int32_t* __restrict__ cnt, int C) {
    __shared__ int32_t part[1024];
    const int t = threadIdx.x;
    const int per = (C + 1023) / 1024;
    const int lo = t * per, hi = min(C, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += cnt[i + 1];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = t ? part[t - 1] : 0;
    if (t == 0) cnt[0] = 0;
    for (int i = lo; i < hi; ++i) {
        run += cnt[i + 1];
        cnt[i + 1] = run;
    }
}

// fill col/val of each row in ascending column order
__global__ __launch_bounds__(256) void csr_fill_kernel(const float* __restrict__ M, int C,
                                                       const int32_t* __restrict__ row_ptr,
                                                       int32_t* __restrict__ col, float* __restrict__ val) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= C) return;
    const float* row = M + (size_t)i * C;
    int base = row_ptr[i];
    for (int j0 = 0; j0 < C; j0 += 64) {
        const int j = j0 + lane;
        const float v = j < C ? row[j] : 0.0f;
        const bool nz = v != 0.0f;
        const unsigned long long m = __ballot(nz);
        if (nz) {
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            col[pos] = j;
            val[pos] = v;
        }
        base += __popcll(m);
    }
}

// Y[i, :] = act(sum_p val[p] * X[col[p], :]) over the CSR row i.  One wave per row, walking the rows grid-stride.
// The row index is wave-uniform, so row_ptr / col / val come through the scalar cache (s_load, no vector-memory
// round trip and no VGPRs) and only the row segments of X are vector loads: 64 lanes x 16 B = one 1-KiB segment
// per (non-zero, 256-feature chunk).  A wave keeps up to 4 non-zeros x 4 chunks = 16 independent segment loads in
// flight (the rows of a PMI-like graph have ~4 non-zeros; a dependent row_ptr -> col -> X chain per 256 features,
// as a (row, chunk) work unit has it, is latency bound).  Non-zeros are accumulated in ascending order (fmaf chain).
template <int NCH>
__device__ __forceinline__ void spmm_row(const int32_t* __restrict__ col, const float* __restrict__ val,
                                         const float* __restrict__ X, int F, int f0, int lo, int hi, int row,
                                         float* __restrict__ Y, int act, int lane, const float* __restrict__ bias) {
    f32x4 acc[NCH];
    bool on[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        on[c] = f0 + c * 256 + lane * 4 < F;
    }
    const float* xb = X + f0 + lane * 4;
    int p = lo;
    for (; p + 4 <= hi; p += 4) {
        const int c0 = col[p], c1 = col[p + 1], c2 = col[p + 2], c3 = col[p + 3];
        const float w0 = val[p], w1 = val[p + 1], w2 = val[p + 2], w3 = val[p + 3];
        f32x4 x[4][NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            x[0][c] = x[1][c] = x[2][c] = x[3][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (on[c]) {
                x[0][c] = *reinterpret_cast<const f32x4*>(xb + (size_t)c0 * F + c * 256);
                x[1][c] = *reinterpret_cast<const f32x4*>(xb + (size_t)c1 * F + c * 256);
                x[2][c] = *reinterpret_cast<const f32x4*>(xb + (size_t)c2 * F + c * 256);
                x[3][c] = *reinterpret_cast<const f32x4*>(xb + (size_t)c3 * F + c * 256);
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[c][j] = fmaf(w0, x[0][c][j], acc[c][j]);
                acc[c][j] = fmaf(w1, x[1][c][j], acc[c][j]);
                acc[c][j] = fmaf(w2, x[2][c][j], acc[c][j]);
                acc[c][j] = fmaf(w3, x[3][c][j], acc[c][j]);
            }
    }
    for (; p < hi; ++p) {
        const int c0 = col[p];
        const float w = val[p];
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (on[c]) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(xb + (size_t)c0 * F + c * 256);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[c][j] = fmaf(w, x[j], acc[c][j]);
            }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (on[c]) {
            // GraphConvolution(bias=True): output + bias behind the product (MODEL:55-56), in front of the activation
            if (bias) acc[c] += *reinterpret_cast<const f32x4*>(bias + f0 + c * 256 + lane * 4);
            const f32x4 o = {mg_act(acc[c][0], act), mg_act(acc[c][1], act), mg_act(acc[c][2], act), mg_act(acc[c][3], act)};
            __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(Y + (size_t)row * F + f0 + c * 256 + lane * 4));
        }
}

__global__ __launch_bounds__(256) void spmm_csr_kernel(const int32_t* __restrict__ row_ptr,
                                                       const int32_t* __restrict__ col,
                                                       const float* __restrict__ val,
                                                       const float* __restrict__ X, int n_rows, int F,
                                                       float* __restrict__ Y, int act, const float* __restrict__ bias) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int nwaves = gridDim.x * 4;
    for (int row = wave; row < n_rows; row += nwaves) {
        const int lo = row_ptr[row], hi = row_ptr[row + 1];
        int f0 = 0;
        for (; f0 + 1024 <= F || (f0 < F && F - f0 > 512); f0 += 1024)
            spmm_row<4>(col, val, X, F, f0, lo, hi, row, Y, act, lane, bias);
        if (f0 < F) {
            if (F - f0 > 256) spmm_row<2>(col, val, X, F, f0, lo, hi, row, Y, act, lane, bias);
            else spmm_row<1>(col, val, X, F, f0, lo, hi, row, Y, act, lane, bias);
        }
    }
}

// Large graphs (X = [n_rows, F] well beyond one XCD's 4-MiB L2): the same product walked in feature SLABS of 64
// floats.  A slab of X is n_rows x 256 B (2.5 MB at 10 000 nodes) and stays resident in the L2 of the XCD that works
// on it, so every gather after the first touch of a row segment is an L2 hit and X crosses the fabric once instead of
// once per non-zero.  Workgroups are dispatched round-robin over the 8 XCDs (blockIdx & 7 = XCD), XCD x owns the adjacent
// slabs [x * spx, (x + 1) * spx).  A wave handles four rows at a time (16 lanes x 16 B = one 256-B row segment each), up to four
// gathers per lane in flight, non-zeros in ascending order exactly like spmm_csr_kernel.
// LPR lanes x 16 B = one row segment (slab width 4 * LPR floats); a pass walks NS adjacent slabs of the XCD at once with
// ONE fetch of the row's col / val for all of them; RU independent rows per lane group.
// What bounds it cache-cold (tools/dev/spmm_exp2.py, N = 10 000, F = 1024, a copy of the same bytes: 17.4 us = 4.70 TB/s):
// the identity graph 16.8 us, a random permutation (one random gather per row) 17.7 us -- the kernel streams at copy speed
// and random first touches cost nothing extra; four gathers per row inside a band (re-reads hit the CU's L1) 19.8 us; four
// RANDOM gathers per row 24.8 us, the 4e-4 random graph (Poisson row lengths) 27.3 us: the loss is the ~3 re-reads of every
// X row, which are L2 / Infinity-Cache hits, not free ones (X = 41 MB does not fit the 32 MB of L2 on the chip).
template <int LPR, int NS, int RU>
__global__ __launch_bounds__(256) void spmm_csr_slab_kernel(const int32_t* __restrict__ row_ptr,
                                                            const int32_t* __restrict__ col,
                                                            const float* __restrict__ val,
                                                            const float* __restrict__ X, int n_rows, int F,
                                                            float* __restrict__ Y, int act, const float* __restrict__ bias) {
    constexpr int SW = 4 * LPR;                 // slab width in floats
    constexpr int GPW = 64 / LPR;               // row groups per wave
    const int xcd = blockIdx.x & 7, bj = blockIdx.x >> 3, nb = gridDim.x >> 3;
    const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
    const int wave = bj * 4 + (threadIdx.x >> 6), nwaves = nb * 4;
    const int nslabs = (F + SW - 1) / SW;
    const int spx = (nslabs + 7) / 8;           // slabs per XCD
    // XCD x owns the ADJACENT slabs [x * spx, (x + 1) * spx): the NS slabs of a pass are one contiguous NS * 256-B piece of
    // every gathered row.  (Round 1 gave slab s to XCD s & 7, i.e. pieces 2 KB apart: 29.1 / 55.9 us cache-cold at N = 10 000,
    // F = 1024 / 2048 against 27.3 / 50.9 us with adjacent slabs, two per pass and 384 workgroups per XCD.)
    for (int slab0 = xcd * spx; slab0 < (xcd + 1) * spx && slab0 < nslabs; slab0 += NS) {
        int f[NS];
        bool fon[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const int sl = slab0 + q;
            f[q] = sl * SW + l * 4;
            fon[q] = sl < nslabs && sl < (xcd + 1) * spx && f[q] < F;
        }
        for (int r0 = wave * GPW * RU; r0 < n_rows; r0 += nwaves * GPW * RU) {
            int p[RU], hi[RU], row[RU];
            f32x4 acc[RU][NS];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                row[u] = r0 + GPW * u + g;
                p[u] = hi[u] = 0;
#pragma unroll
                for (int q = 0; q < NS; ++q) acc[u][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (fon[0] && row[u] < n_rows) {
                    p[u] = row_ptr[row[u]];
                    hi[u] = row_ptr[row[u] + 1];
                }
            }
            bool more = false;
#pragma unroll
            for (int u = 0; u < RU; ++u) more |= p[u] < hi[u];
            while (__any(more)) {
                int c[RU][4];
                float w[RU][4];
                f32x4 x[RU][4][NS];
#pragma unroll
                for (int u = 0; u < RU; ++u)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool live = p[u] + k < hi[u];
                        c[u][k] = live ? col[p[u] + k] : 0;
                        w[u][k] = live ? val[p[u] + k] : 0.f;
                    }
#pragma unroll
                for (int u = 0; u < RU; ++u)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int q = 0; q < NS; ++q) {
                            x[u][k][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (p[u] + k < hi[u] && fon[q]) x[u][k][q] = *reinterpret_cast<const f32x4*>(X + (size_t)c[u][k] * F + f[q]);
                        }
                more = false;
#pragma unroll
                for (int u = 0; u < RU; ++u) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (p[u] + k < hi[u]) {
#pragma unroll
                            for (int q = 0; q < NS; ++q)
#pragma unroll
                                for (int j = 0; j < 4; ++j) acc[u][q][j] = fmaf(w[u][k], x[u][k][q][j], acc[u][q][j]);
                        }
                    p[u] += 4;
                    more |= p[u] < hi[u];
                }
            }
#pragma unroll
            for (int u = 0; u < RU; ++u)
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    if (fon[q] && row[u] < n_rows) {
                        if (bias) acc[u][q] += *reinterpret_cast<const f32x4*>(bias + f[q]);
                        const f32x4 o = {mg_act(acc[u][q][0], act), mg_act(acc[u][q][1], act), mg_act(acc[u][q][2], act),
                                         mg_act(acc[u][q][3], act)};
                        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(Y + (size_t)row[u] * F + f[q]));
                    }
        }
    }
}

}  // namespace

extern "C" int mgnns_gen_adj(const float* A, int C, float* adj, float* work, int32_t* csr_row_ptr,
                             int32_t* csr_col, float* csr_val, mgnns_stream_t stream) {
    MG_REQUIRE(A && adj && work, "mgnns_gen_adj: null pointer");
    MG_REQUIRE(C > 0, "mgnns_gen_adj: C=%d", C);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(rowsum_rsqrt_kernel, dim3((C + 3) / 4), dim3(256), 0, s, A, C, work);
    dim3 tg((C + 31) / 32, (C + 31) / 32);
    hipLaunchKernelGGL(gen_adj_kernel, tg, dim3(256), 0, s, A, C, (const float*)work, adj);
    if (csr_row_ptr) {
        MG_REQUIRE(csr_col && csr_val, "mgnns_gen_adj: csr_col/csr_val required with csr_row_ptr");
        hipLaunchKernelGGL(row_nnz_kernel, dim3((C + 3) / 4), dim3(256), 0, s, (const float*)adj, C, csr_row_ptr);
        hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, csr_row_ptr, C);
        hipLaunchKernelGGL(csr_fill_kernel, dim3((C + 3) / 4), dim3(256), 0, s, (const float*)adj, C,
                           (const int32_t*)csr_row_ptr, csr_col, csr_val);
    }
    MG_CHECK_LAUNCH("mgnns_gen_adj");
    return 0;
}

extern "C" int mgnns_dense_to_csr(const float* M, int C, int32_t* csr_row_ptr, int32_t* csr_col, float* csr_val,
                                  mgnns_stream_t stream) {
    MG_REQUIRE(M && csr_row_ptr && csr_col && csr_val, "mgnns_dense_to_csr: null pointer");
    MG_REQUIRE(C > 0, "mgnns_dense_to_csr: C=%d", C);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(row_nnz_kernel, dim3((C + 3) / 4), dim3(256), 0, s, M, C, csr_row_ptr);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, csr_row_ptr, C);
    hipLaunchKernelGGL(csr_fill_kernel, dim3((C + 3) / 4), dim3(256), 0, s, M, C, (const int32_t*)csr_row_ptr, csr_col,
                       csr_val);
    MG_CHECK_LAUNCH("mgnns_dense_to_csr");
    return 0;
}

extern "C" int mgnns_spmm_csr_bias_fwd(const int32_t* row_ptr, const int32_t* col, const float* val, int n_rows,
                                       const float* X, int F, const float* bias, float* Y, int act, mgnns_stream_t stream) {
    MG_REQUIRE(row_ptr && col && val && X && Y, "mgnns_spmm_csr_fwd: null pointer");
    MG_REQUIRE(!bias || mg_aligned16(bias), "mgnns_spmm_csr_bias_fwd: bias must be 16-byte aligned");
    MG_REQUIRE(n_rows >= 0 && F > 0 && F % 4 == 0, "mgnns_spmm_csr_fwd: F=%d must be a positive multiple of 4", F);
    MG_REQUIRE(mg_aligned16(X) && mg_aligned16(Y), "mgnns_spmm_csr_fwd: X/Y must be 16-byte aligned");
    MG_REQUIRE(act >= 0 && act <= 2, "mgnns_spmm_csr_fwd: unknown activation %d", act);
    if (n_rows == 0) return 0;
    if (F % 4 == 0 && (size_t)n_rows * F * sizeof(float) > ((size_t)6 << 20) && (size_t)n_rows * 256 <= ((size_t)3 << 20)) {
        // X does not fit an XCD's L2 but a 64-float slab of it does: slab-resident walk
        // slabs per XCD walked in ONE pass share a single fetch of the row's col / val (measured at 10 000 nodes: two slabs
        // per pass 20.1 us vs 24.2 us one at a time at F = 1024; four per pass 46.9 us vs 48.9 us at F = 2048)
        const int spx = ((F + 63) / 64 + 7) / 8;
        // two adjacent slabs per pass (512 contiguous bytes of every gathered row), 384 workgroups per XCD (12 per CU: the
        // gathers are latency bound, occupancy is what keeps bytes in flight); four slabs per pass or two rows per lane
        // group were slower cache-cold (F = 2048: 53.1 / 57.9 us against 50.9)
        const dim3 blk(256);
        hipStream_t st = (hipStream_t)stream;
        if (spx >= 2)
            hipLaunchKernelGGL((spmm_csr_slab_kernel<16, 2, 1>), dim3(8 * 384), blk, 0, st, row_ptr, col, val, X, n_rows, F, Y, act, bias);
        else
            hipLaunchKernelGGL((spmm_csr_slab_kernel<16, 1, 2>), dim3(8 * 192), blk, 0, st, row_ptr, col, val, X, n_rows, F, Y, act, bias);
        MG_CHECK_LAUNCH("mgnns_spmm_csr_fwd");
        return 0;
    }
    long long blocks = ((long long)n_rows + 3) / 4;    // one wave per row
    if (blocks > 256 * 8) blocks = 256 * 8;            // 8 workgroups (32 waves) per CU, grid-stride beyond
    hipLaunchKernelGGL(spmm_csr_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, row_ptr, col, val, X,
                       n_rows, F, Y, act, bias);
    MG_CHECK_LAUNCH("mgnns_spmm_csr_fwd");
    return 0;
}

extern "C" int mgnns_spmm_csr_fwd(const int32_t* row_ptr, const int32_t* col, const float* val, int n_rows,
                                  const float* X, int F, float* Y, int act, mgnns_stream_t stream) {
    return mgnns_spmm_csr_bias_fwd(row_ptr, col, val, n_rows, X, F, nullptr, Y, act, stream);
}
