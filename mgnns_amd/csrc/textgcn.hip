// Text-level GCN channel: graph construction + PMI edge-weight lookup + max-times aggregation +
// sum read-out + ReLU in one kernel, one 16-wave workgroup per document (Text_GCN.py:142-275).
//
// Semantics restated: nodes = distinct non-PAD ids of the document; for every position i and every position j within
// +-ngram of it there is an edge tok[i] -> tok[j] (the window includes i itself, and Text_GCN.py:162-164 adds the self loop
// once more) with weight seq_edge_w[edges_matrix[tok[i], tok[j]]] -- id 0 ("no PMI entry") is a learned weight like any
// other; h'_v = max over in-edges of w * h_src; out = relu(sum_v h'_v).
//
// Data layout in HBM: tok [B,T] int64; node_hidden [V,D] rows of 1200 B (D=300) gathered as 16-B lanes; PMI map as CSR
// (row_ptr / col sorted / eid int32; eid == NULL means "id = position + 1", the row-major numbering utils/pmi.py:86-97
// produces), so a lookup is row_ptr -> a window of <= 9 candidate columns fetched at once -> weight: three dependent
// loads (rows average 8 entries; longer rows are bisected down to the window first).
// LDS: the document's node rows staged once (position-major, 16-B lanes), the (2*ngram+1)-wide band of edge weights,
// the compacted token list, same-token chains, per-chunk partial sums.
// Work split: the 16 waves share (a) the row gather, (b) one lookup per thread (n*W <= 1024 for the default shapes),
// (c) the O(n^2) same-token pairing through LDS atomics, and (d) the aggregation as (position chunk x 4-feature group)
// with 16-B LDS reads and the whole window in flight per destination -- the serial per-feature walk over all positions
// of the earlier kernel was ~45 of its 88 us on a 100-token document.
#include "common.hpp"

namespace {

constexpr int TG_MAX_CHUNKS = 16;
constexpr int TG_SHORT_THREADS = 256;
constexpr int TG_SHORT_CAP = 24;      // tokens: 24 x 1200 B of node rows + band + chunk sums = 33 KB of LDS, four workgroups per CU
constexpr int TG_WIN = 8;            // candidate window of a lookup: positions lo .. lo+8

__device__ __forceinline__ float pmi_weight(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                            const int32_t* __restrict__ eid, const float* __restrict__ edge_w,
                                            int n_edge_w, int u, int v) {
    int lo = row_ptr[u];
    const int end = row_ptr[u + 1];
    int hi = end;
    while (hi - lo > TG_WIN) {                      // lower bound stays inside [lo, hi]
        const int mid = (lo + hi) >> 1;
        if (col[mid] < v) lo = mid + 1; else hi = mid;
    }
    int pos = -1;
    const int last = end - 1;
    if (last >= lo) {
        int c[TG_WIN + 1];
#pragma unroll
        for (int t = 0; t <= TG_WIN; ++t) c[t] = col[min(lo + t, last)];     // all candidates in flight together
#pragma unroll
        for (int t = 0; t <= TG_WIN; ++t)
            if (c[t] == v) pos = min(lo + t, last);
    }
    int id = pos < 0 ? 0 : (eid ? eid[pos] : pos + 1);
    id = (id < 0 || id >= n_edge_w) ? 0 : id;
    return edge_w[id];
}

// GT: compile-time ngram (window fully unrolled), 0 = runtime ngram.  NT: threads of the workgroup.
// Two launch forms.  One launch of 1024-thread workgroups with LDS for Tm node rows (120 KB at T = 100): a whole CU per
// document, although a document is a chain of ~5 dependent global round trips that a CU mostly waits for.  Or TWO launches
// (batches from 64 documents on): 256-thread workgroups with LDS for `cap` node rows -- four of them share a CU -- take the
// documents with at most `cap` distinct-position tokens (mode 1: longer ones leave after the token compaction), then the
// 1024-thread form takes the longer ones (mode 2: the short ones leave).  Tn = rows of LDS (Tm or cap).
template <int GT, int NT>
__global__ __launch_bounds__(NT) void textgcn_kernel(const int64_t* __restrict__ tok, int T, int Tm,
                               const float* __restrict__ node_hidden, int V, int D,
                               const float* __restrict__ edge_w, int n_edge_w,
                               const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                               const int32_t* __restrict__ eid, int g_rt, float* __restrict__ out, int vec, int C, int Tn,
                               int mode, int cap) {
    constexpr int TG_THREADS = NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int g = GT > 0 ? GT : g_rt;
    const int W = 2 * g + 1;
    const int D4 = (D + 3) >> 2, Dp = D4 << 2;
    float* s_h = smem;                                   // [Tn][Dp]
    float* s_part = s_h + (size_t)Tn * Dp;               // [C][Dp]
    float* s_w = s_part + (size_t)C * Dp;                // [Tn][W]
    int* s_tok = reinterpret_cast<int*>(s_w + (size_t)Tn * W);   // [Tm]
    int* s_next = s_tok + Tm;                            // [Tm] next position holding the same token, INT_MAX = none
    int* s_first = s_next + Tm;                          // [Tm] 1 if first occurrence of its token
    int* s_n = s_first + Tm;                             // [1]

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
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
    for (int j = tid; j < Tm; j += TG_THREADS) { s_first[j] = 1; s_next[j] = 0x7fffffff; }
    __syncthreads();
    const int n = *s_n;
    if ((mode == 1 && n > cap) || (mode == 2 && n <= cap)) return;         // the other launch's document (uniform per workgroup)

    // -- 2a. stage node rows h[t_i] into LDS: batches of 4 independent 16-B loads per thread ---------------
    if (vec) {
        const int total = n * D4;
        for (int base = tid; base < total; base += 4 * TG_THREADS) {
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = base + k * TG_THREADS;
                if (idx < total) {
                    const int i = idx / D4, c = idx - i * D4;
                    v[k] = *reinterpret_cast<const f32x4*>(node_hidden + (size_t)s_tok[i] * D + 4 * c);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = base + k * TG_THREADS;
                if (idx < total) *reinterpret_cast<f32x4*>(s_h + (size_t)idx * 4) = v[k];      // [i][c] == idx: Dp = 4*D4
            }
        }
    } else {
        for (int idx = tid; idx < n * Dp; idx += TG_THREADS) {
            const int i = idx / Dp, c = idx - i * Dp;
            s_h[idx] = c < D ? node_hidden[(size_t)s_tok[i] * D + c] : 0.f;
        }
    }
    // -- 2b. band of edge weights: s_w[j][o] = edge_w[pmi(t_i, t_j)], i = j - g + o (src i -> dst j) ---
    for (int e = tid; e < n * W; e += TG_THREADS) {
        const int j = e / W, o = e - j * W;
        const int i = j - g + o;
        s_w[e] = (i >= 0 && i < n) ? pmi_weight(row_ptr, col, eid, edge_w, n_edge_w, s_tok[i], s_tok[j]) : 0.f;
    }
    // -- 2c. same-token chains: nodes of the graph are the DISTINCT ids (Text_GCN.py:172).  All pairs k < j. ---------
    for (int pidx = tid; pidx < n * n; pidx += TG_THREADS) {
        const int j = pidx / n, k = pidx - j * n;
        if (k < j && s_tok[k] == s_tok[j]) {
            s_first[j] = 0;
            atomicMin(&s_next[k], j);
        }
    }
    __syncthreads();

    // -- 3. h'_v = max over in-edges (w * h_src); out = relu(sum_v h'_v).  thread = (position chunk c, feature group f):
    //       chunk c owns the first occurrences j0 = c, c+C, ...; repeated tokens walk their chain -----------------------
    const int cidx = tid / D4, f = tid - cidx * D4;
    if (cidx < C) {
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        const f32x4* h4 = reinterpret_cast<const f32x4*>(s_h);
        for (int j0 = cidx; j0 < n; j0 += C) {
            if (!s_first[j0]) continue;
            f32x4 mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            for (int j = j0; j < n; j = s_next[j]) {
                const float* wrow = s_w + (size_t)j * W;
                if (GT > 0) {
                    float w[2 * GT + 1];
                    f32x4 h[2 * GT + 1];
#pragma unroll
                    for (int o = 0; o < 2 * GT + 1; ++o) {
                        const int i = min(max(j - GT + o, 0), n - 1);
                        w[o] = wrow[o];
                        h[o] = h4[(size_t)i * D4 + f];
                    }
#pragma unroll
                    for (int o = 0; o < 2 * GT + 1; ++o) {
                        const int i = j - GT + o;
                        if (i >= 0 && i < n) {
                            mx.x = fmaxf(mx.x, w[o] * h[o].x);
                            mx.y = fmaxf(mx.y, w[o] * h[o].y);
                            mx.z = fmaxf(mx.z, w[o] * h[o].z);
                            mx.w = fmaxf(mx.w, w[o] * h[o].w);
                        }
                    }
                } else {
                    const int lo = max(0, j - g), hi = min(n, j + g + 1);
                    for (int i = lo; i < hi; ++i) {
                        const float w = wrow[i - (j - g)];
                        const f32x4 h = h4[(size_t)i * D4 + f];
                        mx.x = fmaxf(mx.x, w * h.x);
                        mx.y = fmaxf(mx.y, w * h.y);
                        mx.z = fmaxf(mx.z, w * h.z);
                        mx.w = fmaxf(mx.w, w * h.w);
                    }
                }
            }
            sum += mx;
        }
        *reinterpret_cast<f32x4*>(s_part + (size_t)cidx * Dp + 4 * f) = sum;
    }
    __syncthreads();
    for (int d = tid; d < D; d += TG_THREADS) {
        float total = 0.f;
        for (int c = 0; c < C; ++c) total += s_part[(size_t)c * Dp + d];        // fixed order: deterministic
        out[(size_t)b * D + d] = fmaxf(total, 0.f);
    }
}

// LEAN form for the documents of a large batch (round 5): 256 threads, <= 64 VGPRs, NO node rows in LDS -- 8.4 KB of LDS at
// T = 100 (band of edge weights, token list, same-token chains, three chunk sums).  The 1024-thread form above keeps a document's
// node rows in LDS (120 KB at T = 100): a workgroup of it -- even one that only EXITS because its document is short -- needs a CU
// with that much LDS free, and next to the image-bank kernels (148 KB of LDS and 448 of a SIMD's 512 registers per workgroup, alive
// for the whole launch) no CU has: the text GCN took 223 us instead of 37 inside a B = 256 forward and the text->place stack
// started ~60 us late (NOTES_r04 11).  This form fits BESIDE a bank workgroup (12 KB of LDS and 64 registers per SIMD are left):
// the aggregation reads the window's node rows straight from L2 (a row is read once per destination whose window holds it: 9 x
// 1200 B per token, all in flight together), one (position chunk, 4-feature group) per thread as above.  Same arithmetic in the
// same order as the LDS forms: bit-identical results.
template <int GT>
__global__ __launch_bounds__(TG_SHORT_THREADS) void textgcn_lean_kernel(const int64_t* __restrict__ tok, int T, int Tm,
                               const float* __restrict__ node_hidden, int V, int D,
                               const float* __restrict__ edge_w, int n_edge_w,
                               const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                               const int32_t* __restrict__ eid, int g_rt, float* __restrict__ out, int C, int mode, int cap) {
    constexpr int TG_THREADS = TG_SHORT_THREADS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int g = GT > 0 ? GT : g_rt;
    const int W = 2 * g + 1;
    const int D4 = D >> 2, Dp = D4 << 2;                 // (D % 4 == 0: the launcher checks)
    float* s_part = smem;                                // [C][Dp]
    float* s_w = s_part + (size_t)C * Dp;                // [Tm][W]
    int* s_tok = reinterpret_cast<int*>(s_w + (size_t)Tm * W);   // [Tm]
    int* s_next = s_tok + Tm;                            // [Tm] next position holding the same token, INT_MAX = none
    int* s_first = s_next + Tm;                          // [Tm] 1 if first occurrence of its token
    int* s_n = s_first + Tm;                             // [1]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;

    if (tid < 64) {                                      // compact the non-PAD ids (Text_GCN.py:147-150), wave 0
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
    for (int j = tid; j < Tm; j += TG_THREADS) { s_first[j] = 1; s_next[j] = 0x7fffffff; }
    __syncthreads();
    const int n = *s_n;
    if ((mode == 1 && n > cap) || (mode == 2 && n <= cap)) return;         // the other launch's document (uniform per workgroup)

    for (int e = tid; e < n * W; e += TG_THREADS) {      // band of edge weights: s_w[j][o] = edge_w[pmi(t_i, t_j)], i = j - g + o
        const int j = e / W, o = e - j * W;
        const int i = j - g + o;
        s_w[e] = (i >= 0 && i < n) ? pmi_weight(row_ptr, col, eid, edge_w, n_edge_w, s_tok[i], s_tok[j]) : 0.f;
    }
    for (int pidx = tid; pidx < n * n; pidx += TG_THREADS) {      // same-token chains (nodes = DISTINCT ids, Text_GCN.py:172)
        const int j = pidx / n, k = pidx - j * n;
        if (k < j && s_tok[k] == s_tok[j]) {
            s_first[j] = 0;
            atomicMin(&s_next[k], j);
        }
    }
    __syncthreads();

    const int cidx = tid / D4, f = tid - cidx * D4;
    if (cidx < C) {
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        // this thread's 4 features of node row t: byte offset (t * D + 4 f) * 4 into a buffer resource over the table -- 32-bit
        // offsets (64-bit pointers for the nine rows in flight put the ngram-4 instance at 66 registers; 64 is what is left
        // beside an image-bank workgroup)
        const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(node_hidden), 0, 0x7fffffff, 0x00027000);
        const int fo = f * 16, rowb = D * 4;
        for (int j0 = cidx; j0 < n; j0 += C) {
            if (!s_first[j0]) continue;
            f32x4 mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            for (int j = j0; j < n; j = s_next[j]) {
                const float* wrow = s_w + (size_t)j * W;
                if (GT > 0) {
                    float w[2 * GT + 1];
                    f32x4 h[2 * GT + 1];
#pragma unroll
                    for (int o = 0; o < 2 * GT + 1; ++o) {
                        const int i = min(max(j - GT + o, 0), n - 1);
                        w[o] = wrow[o];
                        h[o] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hrs, s_tok[i] * rowb + fo, 0, 0));
                    }
#pragma unroll
                    for (int o = 0; o < 2 * GT + 1; ++o) {
                        const int i = j - GT + o;
                        if (i >= 0 && i < n) {
                            mx.x = fmaxf(mx.x, w[o] * h[o].x);
                            mx.y = fmaxf(mx.y, w[o] * h[o].y);
                            mx.z = fmaxf(mx.z, w[o] * h[o].z);
                            mx.w = fmaxf(mx.w, w[o] * h[o].w);
                        }
                    }
                } else {
                    const int lo = max(0, j - g), hi = min(n, j + g + 1);
                    for (int i = lo; i < hi; ++i) {
                        const float w = wrow[i - (j - g)];
                        const f32x4 h = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hrs, s_tok[i] * rowb + fo, 0, 0));
                        mx.x = fmaxf(mx.x, w * h.x);
                        mx.y = fmaxf(mx.y, w * h.y);
                        mx.z = fmaxf(mx.z, w * h.z);
                        mx.w = fmaxf(mx.w, w * h.w);
                    }
                }
            }
            sum += mx;
        }
        *reinterpret_cast<f32x4*>(s_part + (size_t)cidx * Dp + 4 * f) = sum;
    }
    __syncthreads();
    for (int d = tid; d < D; d += TG_THREADS) {
        float total = 0.f;
        for (int c = 0; c < C; ++c) total += s_part[(size_t)c * Dp + d];        // fixed order: deterministic
        out[(size_t)b * D + d] = fmaxf(total, 0.f);
    }
}

template <int GT>
int launch_lean(const int64_t* tok, int B, int T, int Tm, const float* node_hidden, int V, int D, const float* edge_w, int n_edge_w,
                const int32_t* rp, const int32_t* col, const int32_t* eid, int ngram, float* out, int C, int mode, int cap, size_t lds,
                hipStream_t st) {
    MG_DYN_LDS((textgcn_lean_kernel<GT>), 12 * 1024);
    hipLaunchKernelGGL((textgcn_lean_kernel<GT>), dim3(B), dim3(TG_SHORT_THREADS), lds, st, tok, T, Tm, node_hidden, V, D, edge_w, n_edge_w,
                       rp, col, eid, ngram, out, C, mode, cap);
    MG_CHECK_LAUNCH("mgnns_textgcn_fwd");
    return 0;
}

template <int GT, int NT>
int launch(const int64_t* tok, int B, int T, int Tm, const float* node_hidden, int V, int D, const float* edge_w, int n_edge_w,
           const int32_t* rp, const int32_t* col, const int32_t* eid, int ngram, float* out, int vec, int C, int Tn, int mode, int cap,
           size_t lds, hipStream_t st) {
    MG_DYN_LDS((textgcn_kernel<GT, NT>), NT == 1024 ? 160 * 1024 : 40 * 1024);
    hipLaunchKernelGGL((textgcn_kernel<GT, NT>), dim3(B), dim3(NT), lds, st, tok, T, Tm, node_hidden, V, D, edge_w, n_edge_w,
                       rp, col, eid, ngram, out, vec, C, Tn, mode, cap);
    MG_CHECK_LAUNCH("mgnns_textgcn_fwd");
    return 0;
}

}  // namespace

static int g_textgcn_form = 0;           // 0: by batch and shape; 1 one launch (1024 threads), 2 two launches (short + long), 3 lean
extern "C" int mgnns_textgcn_set_form(int form) {
    MG_REQUIRE(form >= 0 && form <= 3, "mgnns_textgcn_set_form: form=%d (0 by batch, 1 one launch, 2 two launches, 3 lean)", form);
    g_textgcn_form = form;
    return 0;
}

extern "C" int mgnns_textgcn_fwd(const int64_t* tok, int B, int T, const float* node_hidden, int V, int D,
                                 const float* edge_w, int n_edge_w, const int32_t* pmi_row_ptr,
                                 const int32_t* pmi_col, const int32_t* pmi_eid, int ngram, int max_length,
                                 float* out, mgnns_stream_t stream) {
    MG_REQUIRE(B >= 0 && T > 0 && V > 0 && n_edge_w > 0, "mgnns_textgcn_fwd: bad dims B=%d T=%d V=%d", B, T, V);
    if (B == 0) return 0;                       // empty batch: tok / out may be null
    MG_REQUIRE(tok && node_hidden && edge_w && pmi_row_ptr && pmi_col && out, "mgnns_textgcn_fwd: null pointer");
    MG_REQUIRE(D > 0 && D <= 320, "mgnns_textgcn_fwd: D=%d unsupported (1..320)", D);
    MG_REQUIRE(ngram >= 0 && ngram <= 15, "mgnns_textgcn_fwd: ngram=%d unsupported (0..15)", ngram);
    MG_REQUIRE(max_length > 0, "mgnns_textgcn_fwd: max_length=%d", max_length);
    const int Tm = T < max_length ? T : max_length;
    const int W = 2 * ngram + 1;
    const int D4 = (D + 3) / 4, Dp = 4 * D4;
    auto lds_of = [&](int Tn, int C) {
        return ((size_t)Tn * Dp + (size_t)C * Dp + (size_t)Tn * W) * sizeof(float) + (3 * (size_t)Tm + 4) * sizeof(int);
    };
    int C = 1024 / D4;
    C = C > TG_MAX_CHUNKS ? TG_MAX_CHUNKS : C;
    const size_t lds = lds_of(Tm, C);
    MG_REQUIRE(lds <= 160 * 1024, "mgnns_textgcn_fwd: min(T,max_length)=%d needs %zu B of LDS (> 160 KiB)", Tm, lds);
    const int vec = (D % 4 == 0) && mg_aligned16(node_hidden);
    hipStream_t st = (hipStream_t)stream;
    // short documents four to a CU, then the long ones (see the kernel's comment); one launch for small batches, short rows or
    // shapes whose 256-thread form does not fit (MGNNS_TEXTGCN_SPLIT=0: always one launch)
    int Cs = TG_SHORT_THREADS / D4;
    Cs = Cs > TG_MAX_CHUNKS ? TG_MAX_CHUNKS : Cs;
    const size_t lds_s = lds_of(TG_SHORT_CAP, Cs);
    const int form = g_textgcn_form;
    const bool split_ok = Tm > TG_SHORT_CAP && Cs >= 1 && lds_s <= 40 * 1024;
    const bool split = form == 2 ? split_ok : (form == 0 && B >= 64 && split_ok && mg_env_int("MGNNS_TEXTGCN_SPLIT", 1, 7) != 0);
    // large batches (the forward of a B >= 64 batch runs the text GCN next to the image-bank kernels): ONE launch of the LEAN kernel,
    // every document -- the 256-thread short form's 33 KB of LDS does not fit beside a bank workgroup either (12 KB are left), and
    // its launch sat in front of the long documents' in the stream.  MGNNS_TEXTGCN_LEAN=0 or a shape it does not take (D % 4,
    // alignment, a band beyond 12 KB): the two-launch form
    const size_t lds_lean = ((size_t)Cs * Dp + (size_t)Tm * W) * sizeof(float) + (3 * (size_t)Tm + 4) * sizeof(int);
    // (the lean kernel addresses node rows with 32-bit buffer offsets: tables of 2 GiB and more take the 64-bit forms)
    const bool lean_ok = Cs >= 1 && vec && D % 4 == 0 && lds_lean <= 12 * 1024 && (size_t)V * D * 4 < ((size_t)1 << 31);
    const bool lean = form == 3 ? lean_ok : (form == 0 && B >= 64 && lean_ok && mg_env_int("MGNNS_TEXTGCN_LEAN", 1, 8) != 0);
#define TG_ARGS(Tn_, C_, mode_, lds_) tok, B, T, Tm, node_hidden, V, D, edge_w, n_edge_w, pmi_row_ptr, pmi_col, pmi_eid, ngram, out, vec, C_, Tn_, mode_, TG_SHORT_CAP, lds_, st
#define TG_LAUNCH(GT_)                                                                                     \
    {                                                                                                      \
        if (lean)                                                                                          \
            return launch_lean<GT_>(tok, B, T, Tm, node_hidden, V, D, edge_w, n_edge_w, pmi_row_ptr, pmi_col, pmi_eid, ngram, \
                                    out, Cs, 0, TG_SHORT_CAP, lds_lean, st);                               \
        if (split) {                                                                                       \
            if (int rc = launch<GT_, TG_SHORT_THREADS>(TG_ARGS(TG_SHORT_CAP, Cs, 1, lds_s))) return rc;    \
            return launch<GT_, 1024>(TG_ARGS(Tm, C, 2, lds));                                              \
        }                                                                                                  \
        return launch<GT_, 1024>(TG_ARGS(Tm, C, 0, lds));                                                  \
    }
    switch (ngram) {
        case 1: TG_LAUNCH(1)
        case 2: TG_LAUNCH(2)
        case 3: TG_LAUNCH(3)
        case 4: TG_LAUNCH(4)
        default: TG_LAUNCH(0)
    }
#undef TG_LAUNCH
#undef TG_ARGS
}
