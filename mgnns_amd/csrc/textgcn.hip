// Text-level GCN channel: graph construction + PMI edge-weight lookup + max-times aggregation +
// sum read-out + ReLU in one kernel, one workgroup per document (Text_GCN.py:142-275).
//
// Data layout in HBM: tok [B,T] int64; node_hidden [V,D] rows of 1200 B (D=300) read as coalesced
// 16-B lanes; PMI map as CSR (row_ptr/col/eid int32, columns sorted) searched by bisection.
// LDS: the document's node rows h[t_i] are staged once (position-major, n <= Tm rows), the
// (2*ngram+1)-wide band of edge weights, the compacted token list and the same-token chains.
#include "common.hpp"

namespace {

__device__ __forceinline__ int pmi_lookup(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                          const int32_t* __restrict__ eid, int u, int v) {
    int lo = row_ptr[u], hi = row_ptr[u + 1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int c = col[mid];
        if (c < v) lo = mid + 1; else hi = mid;
    }
    return (lo < row_ptr[u + 1] && col[lo] == v) ? eid[lo] : 0;
}

__global__ void textgcn_kernel(const int64_t* __restrict__ tok, int T, int Tm,
                               const float* __restrict__ node_hidden, int V, int D,
                               const float* __restrict__ edge_w, int n_edge_w,
                               const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                               const int32_t* __restrict__ eid, int g, float* __restrict__ out, int vec) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int W = 2 * g + 1;
    float* s_h = smem;                                   // [Tm][D]
    float* s_w = s_h + (size_t)Tm * D;                   // [Tm][W]
    int* s_tok = reinterpret_cast<int*>(s_w + (size_t)Tm * W);   // [Tm]
    int* s_next = s_tok + Tm;                            // [Tm] next position holding the same token, -1 = none
    int* s_first = s_next + Tm;                          // [Tm] 1 if first occurrence of its token
    int* s_n = s_first + Tm;                             // [1]

    const int b = blockIdx.x;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63;

    // -- 1. compact the non-PAD ids (Text_GCN.py:147-150 drops zeros anywhere), wave 0 ----------
    if (tid < 64) {
        int n = 0;
        for (int p0 = 0; p0 < Tm; p0 += 64) {
            const int p = p0 + lane;
            long long id = p < Tm ? tok[(size_t)b * T + p] : 0;
            id = id < 0 ? 0 : (id >= V ? V - 1 : id);
            const bool nz = id != 0;
            const unsigned long long m = __ballot(nz);
            if (nz) s_tok[n + __popcll(m & ((1ull << lane) - 1ull))] = (int)id;
            n += __popcll(m);
        }
        if (lane == 0) *s_n = n;
    }
    __syncthreads();
    const int n = *s_n;

    // -- 2a. stage node rows h[t_i] into LDS --------------------------------------------------------
    if (vec) {
        const int D4 = D >> 2;
        for (int idx = tid; idx < n * D4; idx += nthr) {
            const int i = idx / D4, c = idx - i * D4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(node_hidden + (size_t)s_tok[i] * D + 4 * c);
            *reinterpret_cast<f32x4*>(s_h + (size_t)i * D + 4 * c) = v;
        }
    } else {
        for (int idx = tid; idx < n * D; idx += nthr) {
            const int i = idx / D, c = idx - i * D;
            s_h[(size_t)i * D + c] = node_hidden[(size_t)s_tok[i] * D + c];
        }
    }
    // -- 2b. band of edge weights: s_w[j][o] = edge_w[pmi(t_i, t_j)], i = j - g + o (src i -> dst j) ---
    for (int e = tid; e < n * W; e += nthr) {
        const int j = e / W, o = e - j * W;
        const int i = j - g + o;
        float w = 0.f;
        if (i >= 0 && i < n) {
            int id = pmi_lookup(row_ptr, col, eid, s_tok[i], s_tok[j]);
            id = (id < 0 || id >= n_edge_w) ? 0 : id;
            w = edge_w[id];
        }
        s_w[e] = w;
    }
    // -- 2c. same-token chains: nodes of the graph are the DISTINCT ids (Text_GCN.py:172) ---------------
    for (int j = tid; j < n; j += nthr) {
        const int t = s_tok[j];
        int first = 1;
        for (int k = 0; k < j; ++k)
            if (s_tok[k] == t) { first = 0; break; }
        int nxt = -1;
        for (int k = j + 1; k < n; ++k)
            if (s_tok[k] == t) { nxt = k; break; }
        s_first[j] = first;
        s_next[j] = nxt;
    }
    __syncthreads();

    // -- 3. h'_v = max over in-edges (w * h_src); out = relu(sum_v h'_v); thread = feature dim ----------
    if (tid < D) {
        float sum = 0.f;
        for (int j0 = 0; j0 < n; ++j0) {
            if (!s_first[j0]) continue;
            float mx = -INFINITY;
            for (int j = j0; j >= 0; j = s_next[j]) {
                const int lo = max(0, j - g), hi = min(n, j + g + 1);
                const float* wrow = s_w + (size_t)j * W + (lo - (j - g));
                for (int i = lo; i < hi; ++i) {
                    const float m = wrow[i - lo] * s_h[(size_t)i * D + tid];
                    mx = fmaxf(mx, m);
                }
            }
            sum += mx;
        }
        out[(size_t)b * D + tid] = fmaxf(sum, 0.f);
    }
}

}  // namespace

extern "C" int mgnns_textgcn_fwd(const int64_t* tok, int B, int T, const float* node_hidden, int V, int D,
                                 const float* edge_w, int n_edge_w, const int32_t* pmi_row_ptr,
                                 const int32_t* pmi_col, const int32_t* pmi_eid, int ngram, int max_length,
                                 float* out, mgnns_stream_t stream) {
    MG_REQUIRE(tok && node_hidden && edge_w && pmi_row_ptr && pmi_col && pmi_eid && out,
               "mgnns_textgcn_fwd: null pointer");
    MG_REQUIRE(B >= 0 && T > 0 && V > 0 && n_edge_w > 0, "mgnns_textgcn_fwd: bad dims B=%d T=%d V=%d", B, T, V);
    MG_REQUIRE(D > 0 && D <= 320, "mgnns_textgcn_fwd: D=%d unsupported (1..320)", D);
    MG_REQUIRE(ngram >= 0 && ngram <= 15, "mgnns_textgcn_fwd: ngram=%d unsupported (0..15)", ngram);
    MG_REQUIRE(max_length > 0, "mgnns_textgcn_fwd: max_length=%d", max_length);
    if (B == 0) return 0;
    const int Tm = T < max_length ? T : max_length;
    const int W = 2 * ngram + 1;
    const size_t lds = ((size_t)Tm * D + (size_t)Tm * W) * sizeof(float) + (3 * (size_t)Tm + 4) * sizeof(int);
    MG_REQUIRE(lds <= 160 * 1024, "mgnns_textgcn_fwd: min(T,max_length)=%d needs %zu B of LDS (> 160 KiB)", Tm, lds);
    const int vec = (D % 4 == 0) && mg_aligned16(node_hidden);
    const int threads = ((D + 63) / 64) * 64;
    MG_DYN_LDS(textgcn_kernel, 160 * 1024);
    hipLaunchKernelGGL(textgcn_kernel, dim3(B), dim3(threads), lds, (hipStream_t)stream, tok, T, Tm, node_hidden, V, D,
                       edge_w, n_edge_w, pmi_row_ptr, pmi_col, pmi_eid, ngram, out, vec);
    MG_CHECK_LAUNCH("mgnns_textgcn_fwd");
    return 0;
}
