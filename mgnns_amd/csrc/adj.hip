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

// Y[i, :] = act(sum_p val[p] * X[col[p], :]) over the CSR row i.  Work unit = (row, 256-feature chunk) = one
// wave (64 lanes x 16 B); waves walk the units grid-stride so a CU always has many independent 1-KiB row-segment
// loads in flight (the rows of a PMI-like graph have ~4 non-zeros: a block per row is launch/latency bound).
// The non-zeros of a row are consumed four at a time (four gathers in flight per wave), in ascending order.
__global__ __launch_bounds__(256) void spmm_csr_kernel(const int32_t* __restrict__ row_ptr,
                                                       const int32_t* __restrict__ col,
                                                       const float* __restrict__ val,
                                                       const float* __restrict__ X, int n_rows, int F,
                                                       float* __restrict__ Y, int act) {
    const int lane = threadIdx.x & 63;
    const int chunks = (F + 255) >> 8;
    const long long units = (long long)n_rows * chunks;
    const long long wstride = (long long)gridDim.x * 4;
    for (long long u = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); u < units; u += wstride) {
        const int i = (int)(u / chunks);
        const int f = ((int)(u - (long long)i * chunks) << 8) + lane * 4;
        const int lo = row_ptr[i], hi = row_ptr[i + 1];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (f < F) {
            int p = lo;
            for (; p + 4 <= hi; p += 4) {
                const int c0 = col[p], c1 = col[p + 1], c2 = col[p + 2], c3 = col[p + 3];
                const float w0 = val[p], w1 = val[p + 1], w2 = val[p + 2], w3 = val[p + 3];
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(X + (size_t)c0 * F + f);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(X + (size_t)c1 * F + f);
                const f32x4 x2 = *reinterpret_cast<const f32x4*>(X + (size_t)c2 * F + f);
                const f32x4 x3 = *reinterpret_cast<const f32x4*>(X + (size_t)c3 * F + f);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] = fmaf(w0, x0[j], acc[j]);
                    acc[j] = fmaf(w1, x1[j], acc[j]);
                    acc[j] = fmaf(w2, x2[j], acc[j]);
                    acc[j] = fmaf(w3, x3[j], acc[j]);
                }
            }
            for (; p < hi; ++p) {
                const float w = val[p];
                const f32x4 x = *reinterpret_cast<const f32x4*>(X + (size_t)col[p] * F + f);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(w, x[j], acc[j]);
            }
            f32x4 o = {mg_act(acc[0], act), mg_act(acc[1], act), mg_act(acc[2], act), mg_act(acc[3], act)};
            __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(Y + (size_t)i * F + f));
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

extern "C" int mgnns_spmm_csr_fwd(const int32_t* row_ptr, const int32_t* col, const float* val, int n_rows,
                                  const float* X, int F, float* Y, int act, mgnns_stream_t stream) {
    MG_REQUIRE(row_ptr && col && val && X && Y, "mgnns_spmm_csr_fwd: null pointer");
    MG_REQUIRE(n_rows >= 0 && F > 0 && F % 4 == 0, "mgnns_spmm_csr_fwd: F=%d must be a positive multiple of 4", F);
    MG_REQUIRE(mg_aligned16(X) && mg_aligned16(Y), "mgnns_spmm_csr_fwd: X/Y must be 16-byte aligned");
    MG_REQUIRE(act >= 0 && act <= 2, "mgnns_spmm_csr_fwd: unknown activation %d", act);
    if (n_rows == 0) return 0;
    const long long units = (long long)n_rows * ((F + 255) / 256);
    long long blocks = (units + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;          // 16 workgroups (64 waves) per CU, grid-stride beyond
    hipLaunchKernelGGL(spmm_csr_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, row_ptr, col, val, X,
                       n_rows, F, Y, act);
    MG_CHECK_LAUNCH("mgnns_spmm_csr_fwd");
    return 0;
}
