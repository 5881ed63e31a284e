/*
 * mgnns_hip.h -- C ABI of libmgnns_hip.so: the MI355X (gfx950) forward hot path of MGNNS.
 *
 * The reference has no native code and no FFI: its operators are Python nn.Modules that
 * dispatch to cuDNN/cuBLAS/DGL kernels.  Each entry point below replaces the device work
 * behind one reference call site (cited as file:line into the reference tree) and is what
 * a ctypes binding of that call site would bind.  See INTEGRATION.md for the Python stubs.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (row-major, contiguous);
 *     the library never allocates, frees or synchronises; workspaces are passed in;
 *   - `stream` is a hipStream_t (as void*); all work is enqueued on it, no hidden syncs;
 *   - return value 0 = enqueued, negative = rejected (mgnns_last_error() has the text);
 *   - fp32 everywhere unless an argument says otherwise; int64 token ids as PyTorch makes them.
 */
#ifndef MGNNS_HIP_H
#define MGNNS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mgnns_stream_t;
typedef struct mgnns_comm_s* mgnns_comm_t;      /* an RCCL communicator (one per rank / per device) */

/* activation codes for mgnns_linear_fwd / mgnns_matmul_fwd */
#define MGNNS_ACT_NONE   0
#define MGNNS_ACT_RELU   1   /* F.relu, submodules.py:135 */
#define MGNNS_ACT_LRELU2 2   /* nn.LeakyReLU(0.2), Multi_GCN_Multihead_att.py:306 */

#define MGNNS_ERR_ARG     (-1)
#define MGNNS_ERR_LAUNCH  (-2)
#define MGNNS_ERR_UNSUPP  (-3)

/* Status word of the persistent launches.  mgnns_label_gcn_fwd hands work between the workgroups of ONE launch through an
 * in-order item queue, the fused channel / layer tails exchange partial results inside clusters of adjacent workgroups; none
 * of them needs the whole grid resident, and every wait is bounded: a wait that runs out (a device in trouble, a debugger
 * holding a wave) writes one of the codes below into the registered word and the launch drains instead of hanging.  The next
 * persistent launch of the process returns MGNNS_ERR_LAUNCH with the story in mgnns_last_error() and clears the word;
 * mgnns_take_status() reads-and-clears it explicitly (e.g. after a stream synchronise).
 * host_pinned: 4 bytes of host memory that the device can write (hipHostMalloc / a pinned torch tensor), alive until replaced;
 * NULL unregisters.  Without a registered word the waits are still bounded, only the report is lost. */
#define MGNNS_STATUS_LABEL_GCN_TIMEOUT 1
#define MGNNS_STATUS_CLUSTER_TIMEOUT   2
#define MGNNS_STATUS_BAD_PLAN          3      /* a masked attention launch was handed a plan of another kind / batch: it did nothing */
int mgnns_set_status_word(int32_t* host_pinned);
int mgnns_take_status(void);

/* Text of the last error on the calling thread ("" if none). */
const char* mgnns_last_error(void);
/* ABI version (bumped on any signature change). */
int mgnns_abi_version(void);
/* 16 hex digits: sha256 over the sources this library was built from (every .hip and .hpp file of csrc and every header of
 * include; mgnns_amd/build.py generates the unit).  A measurement records it; the host side refuses to file a profile under
 * sources that differ. */
const char* mgnns_source_fingerprint(void);
/* Measurement, not an operator: does `blockIdx.x & 7` select the XCD on this device / runtime?  Several kernels place work that
 * way (SpMM feature slabs, the dense GEMM's row-block ranges) -- only their SPEED depends on it.  out9[0] = 1 / 0, out9[1 + k] =
 * the hardware XCC_ID observed for block indices with b & 7 == k (-1: more than one).  Synchronises the device. */
int mgnns_xcd_probe(int32_t* out9);

/* ---- a1: text-level GCN channel -------------------------------------------------------
 * Replaces Text_GCN.Model.forward (models/Text_GCN.py:213-275) including the host graph
 * construction seq_to_graph / add_seq_edges (Text_GCN.py:142-211):
 *   per document b: t = non-zero ids of tok[b, :min(T,max_length)];
 *   edge t_i -> t_j for |i-j| <= ngram (and the self loop), weight edge_w[pmi(t_i,t_j)];
 *   h'_v = max_e w_e * node_hidden[src_e];  out[b] = relu(sum over distinct v of h'_v).
 * tok [B,T] int64 (0 = PAD); node_hidden [V,D] (D <= 320); edge_w [n_edge_w] (seq_edge_w
 * [count,1] flattened); PMI map as CSR over V rows with sorted int32 columns and edge ids
 * (id 0 / absent = "no PMI entry", utils/pmi.py:86-97); out [B,D].
 * pmi_eid may be NULL: the id of the entry stored at CSR position k is then k + 1 -- the row-major
 * numbering utils/pmi.py:86-97 hands out -- and the lookup needs one dependent load less.
 * One 1024-thread workgroup per document; needs min(T,max_length)*(D+2*ngram+4)*4 + ~20 KB <= 160 KB of LDS.
 */
int mgnns_textgcn_fwd(const int64_t* tok, int B, int T,
                      const float* node_hidden, int V, int D,
                      const float* edge_w, int n_edge_w,
                      const int32_t* pmi_row_ptr, const int32_t* pmi_col, const int32_t* pmi_eid,
                      int ngram, int max_length, float* out, mgnns_stream_t stream);
/* Launch form of mgnns_textgcn_fwd (results agree to fp32 summation order; a test / measurement knob, process wide):
 * 0 = by batch and shape (default): batches of at least 64 documents run ONE launch of the lean kernel (256 threads, 8 KB of LDS,
 *     <= 64 registers, node rows read from L2: it fits beside an image-bank workgroup of the same forward; also MGNNS_TEXTGCN_LEAN),
 *     smaller ones one launch of 1024-thread workgroups with the document's node rows in LDS;
 * 1 = always that 1024-thread form, 2 = two launches (documents of <= 24 tokens on 256-thread workgroups, then the others on the
 *     1024-thread form; round 3-4's form for large batches), 3 = always the lean kernel (needs D % 4 == 0). */
int mgnns_textgcn_set_form(int form);

/* ---- a2: embedding gather ---------------------------------------------------------------
 * nn.Embedding lookups (Multi_GCN_Multihead_att.py:371; Text_GCN.py:184,206):
 * out[i,:] = table[idx[i],:], idx [n] int64 in [0,V), table [V,D], out [n,D].
 */
int mgnns_embedding_fwd(const int64_t* idx, int64_t n, const float* table, int V, int D,
                        float* out, mgnns_stream_t stream);

/* ---- a2 + a10: text memory bank = embedding gather + packed multi-layer bidirectional LSTM -----------
 * get_text_memory_bank (Multi_GCN_Multihead_att.py:366-398): embedding(text) -> pack_padded_sequence ->
 * nn.LSTM(hidden 150, bidirectional, batch_first) -> pad_packed_sequence(total_length=T).
 * tok [B,T] int64; lens [B] int64 ON THE DEVICE (valid tokens per sample, clamped to [0,T]);
 * emb_table [V,emb_dim]; per layer l: w_ih_cat[l] = [weight_ih_l{l} ; weight_ih_l{l}_reverse] stacked to
 * [2*4*hidden, in_l] and b_ih_cat[l] [2*4*hidden] (both directions' input projections run as one GEMM);
 * per (layer l, direction d) at index 2*l+d (d=1 is "_reverse"): w_hh [4*hidden, hidden], b_hh [4*hidden]
 * (gate order i,f,g,o); all passed as host arrays of device pointers; out [B,T,2*hidden] (zeros at t >= lens[b]);
 * out_bf16 (optional, may be NULL): the same bank as bf16 [B,T,ld_bf16], zero padded -- what
 * mgnns_sq_mha_core_bf16_fwd consumes.
 * workspace: >= mgnns_bilstm_workspace_bytes(B,T,hidden,num_layers) bytes of device memory.
 * hidden == 150, num_layers <= 2.
 */
size_t mgnns_bilstm_workspace_bytes(int B, int T, int hidden, int num_layers);
int mgnns_bilstm_fwd(const int64_t* tok, const int64_t* lens, int B, int T,
                     const float* emb_table, int V, int emb_dim, int hidden, int num_layers,
                     const float* const* w_ih_cat, const float* const* b_ih_cat,
                     const float* const* w_hh, const float* const* b_hh,
                     void* workspace, size_t workspace_bytes, float* out, void* out_bf16, int ld_bf16,
                     mgnns_stream_t stream);
/* Same contract, bf16-mode recurrence: the per-step W_hh . h on v_mfma_f32_4x4x4_16b_bf16 (W_hh and h rounded to bf16 for
 * the product; fp32 accumulation, gates and cell state; fast exp / rcp): ~2e-3 absolute on the bank, 2.5x shorter chain.  The
 * input projections stay on the exact-f32 GEMM.
 * plan_mask / plan (both or neither; round 5): the batch's text mask [B, T] float and mgnns_sq_mha32_plan_ints(B) int32 -- the
 * packing plan of that mask (exactly what mgnns_sq_mha32_plan(plan_mask, B, T, plan) writes) is built by one extra workgroup of
 * this call's first launch, for the two image->text stacks that consume the text bank (Multi_GCN_Multihead_att.py:509-527).
 * Needs B <= 1024, T <= 128, emb_dim % 4 == 0. */
int mgnns_bilstm_bf16_fwd(const int64_t* tok, const int64_t* lens, int B, int T,
                     const float* emb_table, int V, int emb_dim, int hidden, int num_layers,
                     const float* const* w_ih_cat, const float* const* b_ih_cat,
                     const float* const* w_hh, const float* const* b_hh,
                     void* workspace, size_t workspace_bytes, float* out, void* out_bf16, int ld_bf16, const void* prepacked,
                     const float* plan_mask, int32_t* plan, mgnns_stream_t stream);
/* Weights of the bf16 recurrence / projections in their kernel layouts (depends on the weights only: build once per weight
 * version, pass as `prepacked`; NULL = packed on the fly inside every call). */
size_t mgnns_bilstm_bf16_prepack_bytes(int hidden, int num_layers);
int mgnns_bilstm_bf16_prepack(const float* const* w_ih_cat, const float* const* w_hh, int emb_dim, int hidden, int num_layers,
                              void* packed, mgnns_stream_t stream);
/* The layer-0 input projection FOLDED into the embedding table (weights only, once per weight version; MODEL:366-398: the embedding
 * lookup feeds nn.LSTM's first W_ih and nothing else):
 *   table[v, :] = bf16(emb_table[v, :]) . bf16(w_ih_cat0)^T + b_ih_cat0     [V, 2 * 4 * hidden] fp32 (mgnns_bilstm_bf16_table_bytes)
 * by the same bf16 GEMM the forward would run on its gathered rows (same k order per element: bit-identical rows).
 * mgnns_bilstm_bf16_table_fwd = mgnns_bilstm_bf16_fwd reading its layer-0 projection rows out of `gx_table` by token id: no
 * gather / cast of embedding rows and no GEMM in front of the first recurrence.  workspace of the fold:
 * mgnns_bilstm_bf16_fold_workspace_bytes(V) bytes (the bf16 copies of the table and of W_ih). */
size_t mgnns_bilstm_bf16_table_bytes(int V, int hidden);
size_t mgnns_bilstm_bf16_fold_workspace_bytes(int V);
int mgnns_bilstm_bf16_fold_embedding(const float* emb_table, int V, int emb_dim, int hidden, const float* w_ih_cat0,
                                     const float* b_ih_cat0, void* workspace, size_t workspace_bytes, float* table,
                                     mgnns_stream_t stream);
int mgnns_bilstm_bf16_table_fwd(const int64_t* tok, const int64_t* lens, int B, int T,
                     const float* emb_table, int V, int emb_dim, int hidden, int num_layers,
                     const float* const* w_ih_cat, const float* const* b_ih_cat,
                     const float* const* w_hh, const float* const* b_hh,
                     void* workspace, size_t workspace_bytes, float* out, void* out_bf16, int ld_bf16, const void* prepacked,
                     const float* gx_table, const float* plan_mask, int32_t* plan, mgnns_stream_t stream);

/* ---- a3: adjacency normalisation ----------------------------------------------------------
 * gen_adj (utils/util.py:421-426): d = rowsum(A)^-1/2; adj[i,j] = (A[j,i]*d[i])*d[j].
 * A, adj [C,C].  Also emits adj as CSR (row_ptr [C+1] int32, col/val [C*C] capacity,
 * ascending columns) for mgnns_spmm_csr_fwd; csr pointers may be NULL to skip.
 */
int mgnns_gen_adj(const float* A, int C, float* adj, float* work /* [C] floats */,
                  int32_t* csr_row_ptr, int32_t* csr_col, float* csr_val, mgnns_stream_t stream);

/* Dense [C,C] -> CSR (ascending columns; col/val capacity C*C): for GraphConvolution.forward(input, adj)
 * callers that hand over a dense adjacency (MODEL:52). */
int mgnns_dense_to_csr(const float* M, int C, int32_t* csr_row_ptr, int32_t* csr_col, float* csr_val,
                       mgnns_stream_t stream);

/* ---- a4: GraphConvolution = dense X*W then sparse adj*support --------------------------------
 * mgnns_matmul_fwd:   Y[M,N] = act(X[M,K] * W[K,N])      (GraphConvolution.weight layout
 *                     [in,out], Multi_GCN_Multihead_att.py:53)
 * mgnns_spmm_csr_fwd: Y[i,:] = act(sum_p val[p] * X[col[p],:]), p in row i   (MODEL:54 with
 *                     the adjacency in CSR; F % 4 == 0)
 */
int mgnns_matmul_fwd(const float* X, int M, int K, const float* W, int N, float* Y, int act,
                     void* workspace, size_t workspace_bytes, mgnns_stream_t stream);
int mgnns_spmm_csr_fwd(const int32_t* row_ptr, const int32_t* col, const float* val, int n_rows,
                       const float* X, int F, float* Y, int act, mgnns_stream_t stream);
/* GraphConvolution(bias=True) (MODEL:40-41,55-56): Y[i,:] = act(sum_p ... + bias[:]); bias [F] (the reference's
 * [1,1,F] parameter), NULL = the call above. */
int mgnns_spmm_csr_bias_fwd(const int32_t* row_ptr, const int32_t* col, const float* val, int n_rows,
                            const float* X, int F, const float* bias, float* Y, int act, mgnns_stream_t stream);

/* ---- a3/a4 at BASELINE configs[4] scale: bf16-feature sparse propagation ---------------------------------
 * Y[i,:] = act(sum_p val[p] * X[col[p],:]) as above (MODEL:54, adjacency of UTIL:421-426 held sparse) with the
 * adjacency values and X in bf16, fp32 accumulation in ascending column order, Y bf16 (y_bf16 = 1) or fp32.
 * Algorithmic bytes: nnz * (4 + 2) + n_cols * F * 2 + n_rows * F * 2.  X must be finite.
 *
 * mgnns_spmm_csr_bf16_fwd: one gather per non-zero from the XCD's L2 (feature slabs of 128 owned by XCDs); the path for
 *   PMI-like graphs (a few non-zeros per row).  F % 8 == 0; nnz = row_ptr[n_rows]; val_bf16 must be readable up to an
 *   even number of elements (the values travel as aligned dwords).  variant = 0 (default: the pipelined ring form) or a
 *   sweep code selecting one of the three kernel forms and its launch geometry (csrc/spmm_bf16.hip, tools/dev).
 *   row_map (optional, int32 [n_rows]): the CSR's row r is row row_map[r] of Y -- a static adjacency is handed over with its
 *   rows SORTED BY LENGTH (four rows share a wave instruction: with equal lengths nobody idles; ops.SparseAdjBf16 does it),
 *   the values of every row of Y are the same bits either way.  NULL: the CSR's row r is row r of Y.
 * mgnns_spmm_tiled_bf16_fwd: dense-ish graphs (tens of non-zeros per row): X staged through LDS tiles of `tile_cols`
 *   columns by LDS-DMA, a workgroup of 16 waves owns 16 * rows_per_wave rows x 32 * lane_bytes features of Y in
 *   registers.  The adjacency comes as the one-off re-ordered stream of mgnns_amd/spmm_plan.py (format documented
 *   there): wave_off [(ceil(n_rows / (16 rows_per_wave)) * 16) * (ceil(n_cols / tile_cols) + 1)] uint32, ent uint32
 *   (64 dwords of slack at the end).  Built geometries (lane_bytes, rows_per_wave, tile_cols): (8, 10, 128),
 *   (8, 20, 128), (4, 10, 256), (4, 20, 256).  F % (32 lane_bytes) == 0.
 * mgnns_cast_bf16: fp32 -> bf16 (round to nearest even), n elements (adjacency values, features).
 */
int mgnns_cast_bf16(const float* src, long long n, void* dst_bf16, mgnns_stream_t stream);
int mgnns_spmm_csr_bf16_fwd(const int32_t* row_ptr, const int32_t* col, const void* val_bf16, int n_rows, int nnz,
                            const void* X_bf16, int F, void* Y, int y_bf16, int act, int variant,
                            const int32_t* row_map, mgnns_stream_t stream);
int mgnns_spmm_tiled_bf16_fwd(const uint32_t* wave_off, const uint32_t* ent, int lane_bytes, int rows_per_wave,
                              int tile_cols, int n_rows, int n_cols, const void* X_bf16, int F, void* Y, int y_bf16,
                              int act, mgnns_stream_t stream);

/* ---- nn.Linear -------------------------------------------------------------------------------
 * Y[M,N] = act(X[M,K] * W[N,K]^T + bias[N]) (+ residual[M,N]);  bias/residual may be NULL.
 * Serves every nn.Linear / Conv1d(k=1) on the path (MODEL:78-82,320-335; submodules.py:24-26,
 * 34,126-127) and the read-out pooled*G^T (MODEL:474,500) with W = G [C,2048].
 */
int mgnns_linear_fwd(const float* X, int M, int K, const float* W, const float* bias, int N,
                     const float* residual, float* Y, int act,
                     void* workspace, size_t workspace_bytes, mgnns_stream_t stream);
/* Both GEMM entry points split the K dimension over extra workgroups when the 64x64 tiling alone would leave
 * most of the 256 CUs idle (M = batch = 256 layers); `workspace` (device, >= mgnns_gemm_workspace_bytes(), one
 * per stream in flight) receives the partial sums, reduced in a fixed order.  NULL workspace = no K split. */
size_t mgnns_gemm_workspace_bytes(void);

/* ---- a5+a6: image memory bank + global max-pool, one pass over the feature map ------------------
 * get_img_{object,place}_memory_bank (MODEL:400-428) fused with MaxPool2d(14,14) (MODEL:454-455):
 *   bank[b,p,:] = W * feat[b,:,p] + bias    feat [B,K,P] (NCHW map viewed [B,2048,196])
 *   pooled[b,k] = max_p feat[b,k,p]
 * Wt is the Linear weight TRANSPOSED and padded: [K, ldw] with ldw >= N, ldw % 16 == 0, columns
 * N..ldw-1 zero (build it once per weight version with mgnns_transpose_pad).  N <= 304, P <= 208,
 * K % 16 == 0.  bank [B,P,N]; pooled [B,K] (NULL to skip).
 */
int mgnns_imgbank_pool_fwd(const float* feat, int B, int K, int P,
                           const float* Wt, int ldw, const float* bias, int N,
                           float* bank, float* pooled, mgnns_stream_t stream);
/* bf16-operand variant (BASELINE config 3): same reads (the fp32 map crosses HBM once, max-pool exact fp32),
 * W pre-packed by mgnns_imgbank_pack_weights_bf16 into mgnns_imgbank_packed_weight_bytes(K) bytes, the bank is
 * emitted as bf16 [B, P, ld] with ld == 320 (zero padded) -- the layout mgnns_sq_mha_core_bf16_fwd consumes.
 * pooled_work: [B, 2, K] floats: two partial maxima whose max is the pooled value (pooled may be NULL: the caller then
 * consumes these itself, e.g. mgnns_label_tail_fwd with n_parts = 2).  16 <= P <= 208, P % 4 == 0, N <= 304, K % 64 == 0.
 * Two forms, chosen by the batch: one workgroup per sample streaming the map through an LDS-DMA ring (chip-filling batches),
 * two workgroups per sample (region halves; needs 104 < P <= 200, K % 128 == 0) up to half a chip of samples.
 * mgnns_imgbank_set_form: 0 = by batch (default; also MGNNS_IMGBANK_FORM), 1 = always the stream form, 2 = always the pair form
 * where its limits allow -- for tests and measurements; process-wide.
 */
int mgnns_imgbank_set_form(int form);
size_t mgnns_imgbank_packed_weight_bytes(int K);
int mgnns_imgbank_pack_weights_bf16(const float* W, int N, int K, void* Wp, mgnns_stream_t stream);
int mgnns_imgbank_pool_bf16_fwd(const float* feat, int B, int K, int P, const void* Wp, const float* bias, int N,
                                void* bank_bf16, int ld, float* pooled, float* pooled_work, mgnns_stream_t stream);

/* Split-bf16 ("bf16x3") form of the same pass: fp32-class accuracy (every operand as bf16 hi + lo, three MFMAs per product,
 * fp32 accumulation: ~2^-16 relative per product) at bf16-MFMA speed -- the parity-grade mode's image bank.
 * (Wp_hi, Wp_lo) = mgnns_pack_weight_bf16_split of liner_img_*.weight [N,K]; bank [B,P,N] fp32 (+ bias);
 * pooled_halves [B,2,K] fp32 or NULL: exact maxima over the region halves [0,112) / [112,P) (-inf for an empty half).
 * bank_hi / bank_lo (both or neither; round 5): the same bank as split-bf16 images hi = bf16(x), lo = bf16(x - hi), bf16
 * [B,P,320] each, columns [N,320) zero -- what mgnns_sq_mha_core_split_fwd consumes; `bank` may then be NULL (N even).
 * K % 64 == 0, P % 4 == 0, P <= 224, N <= 304. */
int mgnns_imgbank_pool_split_fwd(const float* feat, int B, int K, int P, const void* Wp_hi, const void* Wp_lo,
                                 const float* bias, int N, float* bank, float* pooled_halves, void* bank_hi, void* bank_lo,
                                 mgnns_stream_t stream);

/* out[c, r] = in[r, c] for r < rows, c < cols; out is [cols, ld] with zero padding. */
int mgnns_transpose_pad(const float* in, int rows, int cols, float* out, int ld, mgnns_stream_t stream);

/* ---- a7: label-query attention core -----------------------------------------------------------------
 * The element-wise "attention" of Attention.forward (MODEL:101-131) between the projections:
 *   x[b,l,h*dh+d] = softmax_d(Q[l,h,d]*K[b,h,d]/sqrt(dh)) * V[b,h,d]
 * Q [NLQ,hid], K,V [B,hid] (already through w_q/w_k/w_v), x [B,NLQ,hid]; hid = n_heads*dh, dh <= 64.
 * Replaces the B-iteration torch.cat loop (MODEL:114-115).
 */
int mgnns_label_attn_core_fwd(const float* Q, const float* K, const float* V, int B, int NLQ,
                              int n_heads, int dh, float* x, mgnns_stream_t stream);
/* Attention.forward(mask=...) (MODEL:118-119): energy.masked_fill(mask == 0, -1e10) behind the scaling, in front of
 * the softmax over d.  mask = the byte image [B,NLQ,hid] of the caller's mask broadcast against the energy
 * [B,NLQ,heads,dh] (0 = masked); NULL = the call above. */
int mgnns_label_attn_core_masked_fwd(const float* Q, const float* K, const float* V, const unsigned char* mask,
                                     int B, int NLQ, int n_heads, int dh, float* x, mgnns_stream_t stream);

/* ---- a5 + a7 (fused): everything of an image channel behind the memory-bank kernel, one launch ----------------
 * Read-out (MODEL:454-455 max-pool halves + 474 `matmul(feature, x)`), Attention.forward (MODEL:88-133) without its
 * w_q, then linear_5 and x_linear (MODEL:477-479 / 504-506), optionally the query projection of the fusion stack the
 * result feeds (submodules.py:63-66):
 *   x = max_parts(pooled) . G^T                      (or x given: pass g_wp = NULL)
 *   K = w_k x + b_k, V = w_v x + b_v;  o[b,l] = softmax_d(Q[l] * K[b] / sqrt(dh)) * V[b] per head;
 *   y[b,l] = Wc o[b,l] + bc  with Wc = linear_5.weight . fc.weight [N5,hid], bc = linear_5.weight . fc.bias +
 *   linear_5.bias (the two maps have no non-linearity between them; composed by the caller);
 *   out[b] = x_linear(concat_l y[b,l]);   qh_next[b] = w_qs(out[b]) + b_qs  (wq_next_wp != NULL; needs n_out == hid).
 * pooled [B,n_parts,K_pool] (n_parts = 2: the memory-bank kernel's per-half maxima), g_wp = mgnns_pack_weight_f32
 * image of G [C,K_pool]; Q [NLQ,hid] = w_q(label query); wk_wp / wv_wp / wc_wp / xl_wp / wq_next_wp are
 * mgnns_pack_weight_f32 images of w_k [hid,C], w_v [hid,C], Wc [N5,hid], x_linear.weight [n_out, NLQ*N5],
 * w_qs.weight [HK_next, hid]; out [B,n_out].  hid = n_heads*dh <= 320, dh <= 64, N5 <= 128, NLQ*N5 >= 512 with G.
 * One workgroup per 16 samples; every contraction on the exact-f32 MFMA.
 */
int mgnns_label_tail_fwd(const float* x, int B, int C, const float* pooled, int n_parts, int K_pool,
                         const float* g_wp, const float* Q, int NLQ, int n_heads, int dh,
                         const float* wk_wp, const float* bk, const float* wv_wp, const float* bv,
                         const float* wc_wp, const float* bc, int N5, const float* xl_wp, const float* bxl,
                         int n_out, float* out, const float* wq_next_wp, const float* bq_next, int HK_next,
                         float* qh_next, mgnns_stream_t stream);

/* bf16 precision mode of the same chain, read-out always inside, every contraction on the bf16 MFMA (fp32 accumulation;
 * softmax, biases, attention products fp32).  terms = 1: plain bf16 operands; terms = 3: split-bf16 (hi + lo operands, three
 * MFMAs per product, fp32-class accuracy).  packed[12] = the (hi, lo) buffer pairs of mgnns_pack_weight_bf16_split for
 * G [C,K_pool], w_k, w_v [hid,C], Wc [N5,hid], x_linear.weight [n_out, NLQ*N5], w_qs.weight [HK_next, hid] (last pair may
 * be NULL; lo buffers are not read with terms = 1).  C <= 384, K_pool % 64 == 0, hid <= 320, N5 <= 128, n_out <= 384.
 * cluster_scratch / cluster_counters (both or neither; terms = 3 only): with them FOUR workgroups share each 16-sample
 * tile (read-out K and next-query columns divided, partial read-outs exchanged through the scratch):
 * scratch >= ceil(B/16) * 4 * 24576 B, 16-byte aligned; counters = 2 * ceil(B/16) ints, ZERO before the first launch
 * (every launch leaves them zero).  One (scratch, counters) pair must not be shared by launches that may run concurrently.
 */
int mgnns_label_tail_bf16_fwd(const float* pooled, int B, int n_parts, int K_pool, int C, int terms,
                              const void* const* packed, const float* Q, int NLQ, int n_heads, int dh, const float* bk,
                              const float* bv, const float* bc, int N5, const float* bxl, int n_out, float* out,
                              const float* bq_next, int HK_next, float* qh_next, float* cluster_scratch,
                              int* cluster_counters, mgnns_stream_t stream);

/* ---- a8: single-query multi-head attention, K/V projection fused ---------------------------------------
 * MultiHeadAttention.forward + ScaledDotProductAttention.forward (submodules.py:55-119) for len_q == 1,
 * up to (not including) fc:  per (b,h)
 *   s[l] = qh[b,h,:] . (Wk_h bank[b,l,:] + bk_h) / sqrt(dk);  s[l] = -inf where mask[b,l] == 0
 *   p = softmax_l(s);  o[b,h,:] = sum_l p[l] (Wv_h bank[b,l,:] + bv_h)
 * K and V are never written to memory.  qh [B,H*dk] (already through w_qs); bank [B,L,D];
 * mask [B,L] float or NULL; Wk,Wv [H*dk, D]; o [B,H*dk]; attn [H*B, L] (row h*B+b, submodules.py:72-78)
 * or NULL.  dk == 128, D <= 320, L <= 208.
 */
int mgnns_sq_mha_core_fwd(const float* qh, const float* bank, const float* mask,
                          int B, int L, int D, int H, int dk,
                          const float* Wk, const float* bk, const float* Wv, const float* bv,
                          float* o, float* attn, mgnns_stream_t stream);

/* Head-difference term of MultiHeadAttention(is_regu=True) (models/submodules.py:38-52, 84-93): o [B, H*dv] = the per-head
 * attention outputs (the `o` of mgnns_sq_mha_core_*_fwd) -> out [B] = mean over ordered head pairs i != j of
 * cos^2(o_i, o_j) (n_head == 1: 0 / 0 = NaN, as in the reference).  H <= 16.
 */
int mgnns_head_diff_fwd(const float* o, int B, int H, int dv, float* out, mgnns_stream_t stream);

/* ---- a8, bf16-operand variant (BASELINE config 3: "bf16 MFMA") ------------------------------------------
 * Same contract as mgnns_sq_mha_core_fwd, but the projections run on v_mfma_f32_16x16x32_bf16 (bf16 operands,
 * fp32 accumulation; scores/softmax/weighted sum fp32).  The memory bank is bf16 [B, L, ld] with ld == 320
 * (model dim 300 zero padded; build with mgnns_cast_pad_bf16 or let mgnns_imgbank_pool_bf16_fwd emit it);
 * the K/V weights are pre-packed once per weight version by mgnns_sq_mha_pack_weights_bf16 into
 * mgnns_sq_mha_packed_weight_bytes(H) bytes (MFMA-fragment-major, 1 KiB per fragment).
 */
size_t mgnns_sq_mha_packed_weight_bytes(int H);
int mgnns_sq_mha_pack_weights_bf16(const float* Wk, const float* Wv, int H, int dk, int D, void* Wp,
                                   mgnns_stream_t stream);
int mgnns_cast_pad_bf16(const float* x, int64_t rows, int D, int ld, void* y, mgnns_stream_t stream);
int mgnns_sq_mha_core_bf16_fwd(const float* qh, const void* bank_bf16, const float* mask,
                               int B, int L, int ld, int H, int dk,
                               const void* Wp, const float* bk, const float* bv,
                               float* o, float* attn, mgnns_stream_t stream);

/* ---- a8, bf16 operands on v_mfma_f32_32x32x16_bf16 (round 4) -----------------------------------------------
 * Same contract and inputs as mgnns_sq_mha_core_bf16_fwd (submodules.py:55-119, len_q == 1; moudles.py:207-230 calls it);
 * bank rows in tiles of 32, the model dim in 19 k-steps of 16 (300 -> 304), L <= 224.  The K/V weights are packed by
 * mgnns_sq_mha32_pack_weights_bf16 into mgnns_sq_mha32_packed_weight_bytes(H) bytes (a different fragment order from the
 * 16x16x32 form's).
 * `plan` (optional, needs a mask and L <= 128): the packing plan of the batch's mask -- mgnns_sq_mha32_plan_ints(B) int32 built
 * by mgnns_sq_mha32_plan(mask, B, L, plan) once per batch and shared by every launch on that mask (Multi_GCN_Multihead_att.py:
 * 509-527: both image->text stacks, every layer).  With a plan the live rows of several short samples share a workgroup
 * (8-row aligned, <= 128 rows and <= 16 samples per group): the weight stream is read once per group instead of once per
 * sample.  Same results as without a plan up to fp32 summation order.  plan == NULL: one workgroup per sample.
 */
size_t mgnns_sq_mha32_packed_weight_bytes(int H);
int mgnns_sq_mha32_pack_weights_bf16(const float* Wk, const float* Wv, int H, int dk, int D, void* Wp,
                                     mgnns_stream_t stream);
size_t mgnns_sq_mha32_plan_ints(int B);
int mgnns_sq_mha32_plan(const float* mask, int B, int L, int32_t* plan, mgnns_stream_t stream);
int mgnns_sq_mha32_core_bf16_fwd(const float* qh, const void* bank_bf16, const float* mask,
                                 int B, int L, int ld, int H, int dk,
                                 const void* Wp, const float* bk, const float* bv,
                                 float* o, float* attn, const int32_t* plan, mgnns_stream_t stream);

/* ---- a8, split-bf16 operands ("bf16x3", round 5): the reference's formulation inside the 1e-4 parity gate on the bf16 pipe ----
 * Same contract as mgnns_sq_mha_core_fwd (models/submodules.py:55-119, len_q == 1: K and V projected from the memory bank,
 * scores / mask / softmax / weighted sum fp32, K and V never written to memory).  Every fp32 operand of the two projections is
 * carried as hi = bf16(x), lo = bf16(x - hi) and a product is three v_mfma_f32_16x16x32_bf16 (hi hi + lo hi + hi lo, fp32
 * accumulation, ~2^-16 relative per product).  The memory bank arrives as two bf16 images [B, L, ld], ld == 320 (model dim 300
 * zero padded): build them with mgnns_split_pad_bf16 from the fp32 bank.  The K/V weights are packed once per weight version by
 * mgnns_sq_mha_pack_weights_split into mgnns_sq_mha_split_packed_weight_bytes(H) bytes (hi image, then lo image, each in the
 * fragment-major order of mgnns_sq_mha_pack_weights_bf16).  The bank rows are walked in two halves of at most 112 (the two images
 * of 196 rows do not fit the 160 KB of LDS) joined by an exact fp32 online-softmax merge.  dk == 128, L <= 208, H <= 16.
 */
size_t mgnns_sq_mha_split_packed_weight_bytes(int H);
int mgnns_sq_mha_pack_weights_split(const float* Wk, const float* Wv, int H, int dk, int D, void* Wp,
                                    mgnns_stream_t stream);
int mgnns_split_pad_bf16(const float* x, int64_t rows, int D, int ld, void* hi, void* lo, mgnns_stream_t stream);
int mgnns_sq_mha_core_split_fwd(const float* qh, const void* bank_hi, const void* bank_lo, const float* mask,
                                int B, int L, int ld, int H, int dk,
                                const void* Wp, const float* bk, const float* bv,
                                float* o, float* attn, const int32_t* plan, mgnns_stream_t stream);
/* `plan` (optional: needs a mask, L <= 112, attn == NULL): the group plan of the batch's mask, mgnns_sq_mha32_plan_ints(B) int32 built
 * by mgnns_sq_mha_split_plan(mask, B, L, plan) once per batch (the layout of mgnns_sq_mha32_plan with other constants: samples in
 * whole tiles of 16 rows, at most 112 rows and 7 samples per group).  The samples of a group share one staging of the bank images
 * and one pass of the weight stream (Multi_GCN_Multihead_att.py:509-527: the text bank, mean 16 live rows of 100).  Same results
 * as without a plan up to fp32 summation order. */
int mgnns_sq_mha_split_plan(const float* mask, int B, int L, int32_t* plan, mgnns_stream_t stream);

/* ---- a8, folded variant: the K/V projections folded into the query side ---------------------------------
 * Same inputs and outputs as mgnns_sq_mha_core_fwd (submodules.py:55-119, len_q == 1) computed as
 *   U_h = Wk_h^T qh_h;   p = softmax_l(U_h . bank[b,l,:] / sqrt(dk)) (masked);   o_h = Wv_h (sum_l p_l bank[b,l,:]) + bv_h
 * which is algebraically the reference's result (q.bk is constant over l and drops out of the softmax; sum_l p_l = 1
 * carries bv through) at 1/100 of the FLOPs, all in fp32 (f32 MFMA): the folded attention of the fp32 / bf16x3 modes (the
 * faithful kernels above are what the MFMA-utilisation target is measured on; bf16 mode: mgnns_sq_mha_folded_bf16_fwd below).
 * bank: fp32 [B, L, D] (bank_is_bf16 == 0, ld_bank == D) or bf16 [B, L, ld_bank] zero padded (bank_is_bf16 == 1).
 * workspace: mgnns_sq_mha_folded_workspace_bytes(B, D, H) bytes.  D <= 320, D % 4 == 0, H <= 8, dk % 4 == 0, L <= 208.
 */
size_t mgnns_sq_mha_folded_workspace_bytes(int B, int D, int H);
int mgnns_sq_mha_folded_fwd(const float* qh, const void* bank, int bank_is_bf16, int ld_bank, const float* mask,
                            int B, int L, int D, int H, int dk, const float* Wk, const float* Wv, const float* bv,
                            void* workspace, size_t workspace_bytes, float* o, float* attn, mgnns_stream_t stream);

/* ---- a8, folded variant on the bf16 matrix pipe (bf16 precision mode) ----------------------------------------------
 * The same attention (submodules.py:55-119, len_q == 1) with ALL FOUR projections composed into the maps either side of it
 * (the host mirror builds them once per weight version, fusion.py):
 *   u_h = (Wk_h^T Wq_h) x + Wk_h^T bq_h                                     -- the previous tail's "next projection"
 *   p   = softmax_l(u_h . bank[b,l,:] / sqrt(dk)) (masked);   c_h = sum_l p_l bank[b,l,:]            -- this kernel
 *   fc(o) = sum_h (fc_h Wv_h) c_h + (fc bv + b_fc)                          -- the tail's first map (mgnns_mha_tail_c16_fwd)
 * U: fp32 [B, H*D] (head h at h*D); bank: bf16 [B, L, 320] zero padded; mask: fp32 [B, L] (0 = masked) or NULL;
 * C: bf16 [B, ldc] (ldc >= H*D, ldc % 8 == 0; head h at h*D, zeros behind H*D -- ldc = H*D rounded up to 32 is what
 * mgnns_mha_tail_c16_fwd takes); attn: fp32 [H*B, L] or NULL.  D <= 320, D % 4 == 0, H <= 8, L <= 208.  One read of the bank.
 */
int mgnns_sq_mha_folded_bf16_fwd(const float* U, const void* bank_bf16, const float* mask, int B, int L, int D, int H,
                                 float inv_temp, void* C_bf16, int ldc, float* attn, mgnns_stream_t stream);

/* ---- a8, rest of the layer: fc + residual + LayerNorm + position-wise FFN + residual + LayerNorm in ONE launch ----
 * MultiHeadAttention.forward after the attention (submodules.py:88-94) and PositionwiseFeedForward.forward
 * (submodules.py:132-139), optionally followed by the NEXT layer's query projection w_qs (submodules.py:68):
 *   y = LN1(fc(o) + q);  out = LN2(w_2 relu(w_1 y + b_1) + b_2 + y);  qh_next = w_qs'(out) + b'
 * o [B, HK] (HK = n_head*d_v), q [B, 300] (the layer's query = the residual), out [B, 300], qh_next [B, HK_next].
 * Every weight is passed PRE-PACKED by mgnns_pack_weight_f32 (MFMA-fragment-major fp32, one 16-B load = four
 * k-steps): fc [300, HK], w_1 / w_2 [300, 300] (Conv1d k=1 weight viewed 2-D), w_qs' [HK_next, 300].
 * wq_next_wp == NULL skips the last step.  Exact fp32.  d_model == 300.
 */
size_t mgnns_packed_f32_weight_bytes(int N, int K);
int mgnns_pack_weight_f32(const float* W, int N, int K, float* Wp, mgnns_stream_t stream);
int mgnns_mha_tail_fwd(const float* o, int HK, const float* q, int B, int d_model,
                       const float* fc_wp, const float* fc_b, const float* ln1_gamma, const float* ln1_beta,
                       const float* w1_wp, const float* b1, const float* w2_wp, const float* b2,
                       const float* ln2_gamma, const float* ln2_beta, float eps, float* out,
                       const float* wq_next_wp, const float* bq_next, int HK_next, float* qh_next,
                       mgnns_stream_t stream);

/* bf16-MFMA variant of the fused tail (bf16 precision mode).  terms = 1: bf16 operands; terms = 3: split-bf16
 * (x = hi + lo, a*b ~ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, fp32 accumulate: ~2^-16 relative per product).  Weights
 * pre-packed by mgnns_pack_weight_bf16_split into TWO buffers (hi, lo) of mgnns_packed_bf16_weight_bytes(N,K)
 * bytes each; `packed` = host array of 8 device pointers {fc_hi, fc_lo, w1_hi, w1_lo, w2_hi, w2_lo, wq'_hi, wq'_lo}
 * (the last two NULL to skip the next-layer projection).  Biases, residuals and LayerNorms are fp32.
 */
size_t mgnns_packed_bf16_weight_bytes(int N, int K);
int mgnns_pack_weight_bf16_split(const float* W, int N, int K, void* Whi, void* Wlo, mgnns_stream_t stream);
int mgnns_mha_tail_bf16_fwd(const float* o, int HK, const float* q, int B, int d_model, int terms,
                            const void* const* packed,
                            const float* fc_b, const float* ln1_gamma, const float* ln1_beta, const float* b1,
                            const float* b2, const float* ln2_gamma, const float* ln2_beta, float eps, float* out,
                            const float* bq_next, int HK_next, float* qh_next, int cluster, float* cluster_scratch,
                            int* cluster_counters, mgnns_stream_t stream);
/* cluster / cluster_scratch / cluster_counters as in mgnns_mha_tail_c16_fwd below (terms = 1 only; NULL buffers: every workgroup
 * of a tile's cluster streams all of fc and the ranks share the next layer's w_qs columns): with the buffers the ranks split the K
 * of fc (submodules.py:88), the last arriver finishes the tile and the next layer's w_qs runs as a second launch. */

/* The bf16 tail (terms = 1) behind mgnns_sq_mha_folded_bf16_fwd: `c` = bf16 [B, HC] (HC = n_head * d_model rounded up to a
 * multiple of 32, <= 2560; zeros behind n_head * d_model) instead of the fp32 head outputs; packed[0] = the composed map
 * fc . blockdiag(Wv) [d_model, n_head * d_model], fc_b = fc bv + b_fc; packed[6] / bq_next = the NEXT layer's composed query map
 * [HC_next, d_model] and its bias (NULL: none); only the hi buffers (even entries of `packed`) are read.
 * cluster: workgroups per 16-sample tile (0 = default: 4 with the exchange buffers; 1..8).  With cluster_scratch
 * (mgnns_mha_tail_c16_scratch_floats(B, 8) floats, 16-byte aligned) and cluster_counters (ceil(B/16) int32, ZERO before the
 * first launch; the kernel leaves them zero; one pair of buffers per concurrently running launch) the ranks split the K of the
 * first product, the rank that arrives LAST adds the partial sums and finishes the tile (nobody waits), and the next layer's
 * query map runs as a second launch behind it; without them every rank streams the whole composed map and takes a share of the
 * next one.  Everything else as mgnns_mha_tail_bf16_fwd.
 */
size_t mgnns_mha_tail_c16_scratch_floats(int B, int cluster);
int mgnns_mha_tail_c16_fwd(const void* c_bf16, int HC, const float* q, int B, int d_model, const void* const* packed,
                           const float* fc_b, const float* ln1_gamma, const float* ln1_beta, const float* b1,
                           const float* b2, const float* ln2_gamma, const float* ln2_beta, float eps, float* out,
                           const float* bq_next, int HC_next, float* u_next, int cluster, float* cluster_scratch,
                           int* cluster_counters, mgnns_stream_t stream);

/* ---- a3 + a4 (+ a7's w_q): one image channel's label GCN as ONE persistent launch ---------------------------------
 * gen_adj (utils/util.py:421-426) + GraphConvolution x 2 with LeakyReLU(0.2) between them
 * (Multi_GCN_Multihead_att.py:460-473 / 489-499, 42-58) + optionally w_q(label_query) (MODEL:97):
 *   adj = D^-1/2 A^T D^-1/2;  X1 = lrelu(adj @ (inp @ W1));  G = adj @ (X1 @ W2);  Q = label_query @ wq^T + bq
 * as a grid of `grid` (0: a quarter of the CUs, 64 on an MI355X -- the measured optimum with two channels' launches side by
 * side) workgroups that pull the phases' work items from one in-order queue
 * instead of 11-12 dependent launches.  split = 0: exact-fp32 MFMA, w1a / w2a = mgnns_pack_weight_f32 of W1^T [N1,K0] /
 * W2^T [N2,N1] (bit-equal to mgnns_gen_adj + mgnns_matmul_fwd + mgnns_spmm_csr_fwd); split = 1: split-bf16 operands
 * (fp32-class), (w1a, w1b) / (w2a, w2b) = the (hi, lo) buffers of mgnns_pack_weight_bf16_split.
 * A [C,C]; inp [C,K0]; G [C,N2]; Gp_hi / Gp_lo (both or neither): the mgnns_pack_weight_bf16_split image of G the fused
 * channel tail reads; Q [NLQ,HQ] or NULL.  C <= 512, K0 % 4 == 0, N1, N2 % 256 == 0, N1 <= 1024.
 * scratch: mgnns_label_gcn_scratch_bytes(C, N1, N2) bytes, 256-byte aligned, its first 256 bytes ZERO before the first
 * launch (every launch leaves them zero); one scratch must not be shared by launches that may run concurrently (it holds
 * the launch's intermediates).  The work items of the four phases are pulled in order from one ticket counter, so the
 * launch makes progress with ANY number of its workgroups resident: any grid, any neighbours (see the status word above).
 */
/* Shape limits of the three fused launches as predicates (1 = the launch takes these shapes): the host side routes
 * everything else to the chain of separate operators, which has no such limits.  K_pool = 0 for mgnns_label_tail_supported:
 * the read-out x is passed in. */
int mgnns_label_gcn_supported(int C, int K0, int N1, int N2, int split);
int mgnns_label_tail_supported(int C, int NLQ, int n_heads, int dh, int N5, int n_out, int K_pool, int with_next_q);
int mgnns_label_tail_bf16_supported(int C, int NLQ, int n_heads, int dh, int N5, int n_out, int K_pool, int terms, int with_next_q);
size_t mgnns_label_gcn_scratch_bytes(int C, int N1, int N2);
int mgnns_label_gcn_fwd(const float* A, int C, const float* inp, int K0, int split, const void* w1a, const void* w1b, int N1,
                        const void* w2a, const void* w2b, int N2, float* G, void* Gp_hi, void* Gp_lo,
                        const float* label_query, int NLQ, const float* wq, const float* bq, int HQ, float* Q,
                        void* scratch, size_t scratch_bytes, int grid, mgnns_stream_t stream);

/* ---- a9: classifier over the four fusion features (Multi_GCN_Multihead_att.py:560-566, eval: dropout is the identity) --
 * logits[b,:] = W [f0[b]; f1[b]; f2[b]; f3[b]] + bias without materialising the concatenation.  f_p [B,D]; W [NL, 4*D];
 * bias [NL]; logits [B,NL].  The host passes W = multi_linear_2.weight . multi_linear_1.weight (no non-linearity between
 * them in eval) or any single [NL, 4D] map.
 */
int mgnns_classifier_head_fwd(const float* f0, const float* f1, const float* f2, const float* f3, int B, int D,
                              const float* W, const float* bias, int NL, float* logits, mgnns_stream_t stream);
/* The same classifier as nparts launches, one behind each producer of a feature (the four fusion stacks run on different streams):
 * launch `part` adds  parts[part][b][:] = W[:, part*D:(part+1)*D] . f[b]  and the launch that finishes LAST adds the parts in index
 * order + bias into logits [B, NL] -- no launch (and no stream join) of its own behind the slowest stack.  parts [nparts, B, NL],
 * counter: one int32, zero before the first of the nparts launches (the last launch re-arms it).  All nparts launches must use the
 * same B, D, NL, parts, counter, logits.
 */
int mgnns_classifier_part_fwd(const float* f, int part, int nparts, int B, int D, const float* W, const float* bias,
                              int NL, float* parts, int* counter, float* logits, mgnns_stream_t stream);

/* ---- custom LayerNorm (submodules.py:153-156): unbiased std, eps added to std ---------------------------
 * y[r,:] = gamma * (x[r,:] - mean) / (std_unbiased + eps) + beta,  x,y [rows, D], D <= 1024.
 */
int mgnns_layernorm_fwd(const float* x, int rows, int D, const float* gamma, const float* beta,
                        float eps, float* y, mgnns_stream_t stream);

/* ---- dense bf16 GEMM (BASELINE configs[4] (i): dense [N,N] adjacency x support on the bf16 MFMA; any large X.W) ------
 * C[M,N] = act(A[M,K] . Bt[N,K]^T + bias): A and Bt are bf16 with K-contiguous rows of Kp elements (Kp % 64 == 0, zero
 * padded: build A with mgnns_cast_pad_bf16, Bt from a [K,N] fp32 matrix with mgnns_transpose_cast_bf16), C fp32
 * (c_bf16 = 0) or bf16 (c_bf16 = 1: the K-contiguous operand of the next product, no cast pass) with row stride ldc
 * elements.  N % 4 == 0, ldc % 4 == 0.  UTIL:421-426 (`adj @ support`), MODEL:52-58.
 */
int mgnns_transpose_cast_bf16(const float* x, int rows, int cols, int ld, void* y, mgnns_stream_t stream);
/* workspace (optional, mgnns_gemm_bf16_workspace_bytes() bytes, 16-byte aligned, one per concurrently running launch): with it, a
 * last PARTIAL round of tiles on the persistent workgroups (10 000 x 1024 on 256 compute units: 320 tiles of 256 x 128 = 1.25 rounds)
 * is cut along K over the idle workgroups, partial sums go to the workspace and a second, small launch on the same stream adds a
 * tile's parts in part order (deterministic: no counters, nobody waits) and stores C; and products with K >= 4096 and at least one
 * full round of 256 x 256 tiles per XCD run the 256 x 256-tile kernel (a third less L2 traffic per MAC), whose left-over tiles are
 * cut the same way.  NULL: every tile is computed whole by the 256 x 128 kernel. */
size_t mgnns_gemm_bf16_workspace_bytes(void);
/* c_bf16 (round 6: two bits): bit 0 = C is bf16 (else fp32); bit 1 = TRANSPOSED store: C is C^T [N, ldc >= M] (a product with a small M runs as
 * its transpose and still leaves the K-contiguous operand the next product needs; the 160 x 256 kernel only: ceil(M / 160) >= 8, N >= 256, K >= 320). */
int mgnns_gemm_bf16_nt_fwd(const void* A, const void* Bt, int M, int N, int Kp, const float* bias, void* C, int ldc,
                           int c_bf16, int act, void* workspace, size_t workspace_bytes, mgnns_stream_t stream);
/* Which tile shape mgnns_gemm_bf16_nt_fwd runs (tests and A/B timings; production leaves it alone): -1 the environment
 * (MGNNS_GEMM_160, default 2), 0 round 4's kernels only (256 x 128, 256 x 256), 1 the 160 x 256 kernel whenever the shape fits it,
 * 2 by the launcher's estimate, 3 the 320 x 256 kernel whenever the shape fits it.  100 / 101 / 102: the K-slice width of the 160 x 256 tile --
 * the environment (MGNNS_GEMM160_BK, default 64) / 32 (round 5's kernel: 64-B row segments) / 64 (round 6: whole 128-B lines, half the requests). */
int mgnns_gemm_bf16_set_form(int form);
/* The launcher's choice for a product, by its estimate (host arithmetic, no device call; n_cu = compute units of the device, 256 on
 * MI355X; with_workspace: the workspace of mgnns_gemm_bf16_workspace_bytes() is passed): 4 = 160 x 256 tiles, 5 = 320 x 256 tiles,
 * 0 = round 4's kernels (256 x 128, or 256 x 256 for K >= 4096 with at least a full round of tiles per XCD); < 0 = bad arguments. */
int mgnns_gemm_bf16_pick_form(int M, int N, int Kp, int with_workspace, int n_cu);

/* ---- f4 (metrics half): the evaluation tail after the logits (ENGINE:828-838) -------------------------------------
 * probs = softmax(logits, dim=1) (max-subtracted), pred = first arg-max of probs; when target (int64 [B]) and
 * confusion (int32 [NL, NL], rows = target, columns = prediction; ACCUMULATED, zero it per epoch) are given the batch
 * is added to the confusion matrix, from which accuracy and the micro / macro / weighted F1 of the engine follow
 * (mgnns_amd/metrics.py).  probs, pred, target + confusion may each be NULL.  NL <= 64.
 */
int mgnns_softmax_argmax_fwd(const float* logits, int B, int NL, float* probs, int32_t* pred, const int64_t* target,
                             int32_t* confusion, mgnns_stream_t stream);

/* ---- f4 (trunk half): torchvision-style ResNet trunks cut after layer4 (MODEL:274-294 `object_features` /
 * `place_features`, MODEL:586-595, 629-630), eval mode.  Activations NHWC bf16, fp32 accumulation.
 *  mgnns_conv_fold_bn_bf16   one-off: wt[o, k] = bf16(w[o,c,kh,kw] * gamma[o] / sqrt(var[o] + eps)) with k = (kh, kw, c)
 *                            (k_order 0, rows of ld >= KH*KW*Cin elements, zero padded) or k = (c, kh, kw) (k_order 1: the
 *                            stem, ld = 160); bias[o] = beta[o] + (conv_bias[o] - mean[o]) * scale.  gamma == NULL: no
 *                            BatchNorm (scale 1, bias = conv_bias or 0); conv_bias may be NULL.
 *  mgnns_stem_conv7_fwd      conv1 (7x7, stride 2, pad 3, 3 -> 64) + bn1 + ReLU: img [B,3,H,W] fp32 NCHW ->
 *                            y [B, H/2, W/2, 64] bf16 NHWC (OH = (H - 1) / 2 + 1).
 *  mgnns_maxpool3x3s2_nhwc_fwd  MaxPool2d(3, 2, 1): [B,H,W,C] -> [B,(H-1)/2+1,(W-1)/2+1,C], C % 8 == 0.
 *  mgnns_conv_bf16_nhwc_fwd  y = relu?(conv(x; wt) + bias + residual?) as an implicit GEMM: 1x1 or 3x3, any stride,
 *                            pad <= K/2, Cin a power of two >= 64, Cout % 8 == 0; residual [B,OH,OW,Cout] bf16 or NULL;
 *                            y [B,OH,OW,Cout] bf16, or -- out_nchw_f32 != 0 -- [B,Cout,OH,OW] fp32, the feature-map
 *                            layout mgnns_imgbank_pool_* read.
 */
int mgnns_conv_fold_bn_bf16(const float* w, const float* conv_bias, int Cout, int Cin, int KH, int KW,
                            const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                            int k_order, int ld, void* wt, float* bias, mgnns_stream_t stream);
int mgnns_stem_conv7_fwd(const float* img, int B, int H, int W, const void* wt, const float* bias, void* y,
                         mgnns_stream_t stream);
int mgnns_maxpool3x3s2_nhwc_fwd(const void* x, int B, int H, int W, int C, void* y, mgnns_stream_t stream);
int mgnns_conv_bf16_nhwc_fwd(const void* x, int B, int H, int W, int Cin, const void* wt, const float* bias, int Cout,
                             int KH, int KW, int stride, int pad, const void* residual, int relu, int out_nchw_f32,
                             void* y, mgnns_stream_t stream);

/* ---- a8 (fused layer, bf16 mode): attention core + the rest of the layer in ONE launch ------------------------------
 * mgnns_sq_mha_core_bf16_fwd followed by mgnns_mha_tail_bf16_fwd(terms = 1) for every 16-sample tile, the tail run by the
 * tile's LAST attention-core workgroup to finish (system-scope write-through stores of o, one relaxed agent-scope atomic per
 * workgroup on tile_counters[tile], no fences, nobody spins).  Results are bit-identical to the two separate launches.
 * o_scratch [B, H*dk] fp32 (the cores' output, not needed afterwards); q_in [B, d_model] the layer input (residual);
 * packed[8] as in mgnns_mha_tail_bf16_fwd; tile_counters: ceil(B/16) int32, ZERO before the first launch (the kernel
 * leaves them zero), one array per concurrently running launch.
 */
int mgnns_sq_mha_layer_bf16_fwd(const float* qh, const void* bank_bf16, const float* mask, int B, int L, int ld, int H, int dk,
                                const void* Wp, const float* bk, const float* bv, float* o_scratch, const float* q_in,
                                int d_model, const void* const* packed, const float* fc_b, const float* ln1_gamma,
                                const float* ln1_beta, const float* b1, const float* b2, const float* ln2_gamma,
                                const float* ln2_beta, float eps, float* out, const float* bq_next, int HK_next,
                                float* qh_next, int* tile_counters, mgnns_stream_t stream);

/* ---- measurement aid: a one-thread kernel that writes the GPU's constant-rate real-time counter (s_memrealtime,
 * 100 MHz) into slots[idx] when the stream reaches it.  Captured into the forward's hipGraph it gives the REAL
 * timeline of the concurrent branches of a replay (rocprofv3 serialises / perturbs them): tools/graph_timeline.py.
 */
int mgnns_debug_stamp(uint64_t* slots, int idx, mgnns_stream_t stream);

/* Access-pattern measurement aid (tools/dev/slabcopy_exp.py): copy / read / write a [n_rows, pitch] byte matrix in pieces of
 * `piece` bytes dealt to XCDs the way the slab SpMM kernels deal their row segments (mode: 0 copy, 1 read, 2 write; +16 plain
 * instead of non-temporal stores; +256 ignore XCDs).  wgx = workgroups per XCD. */
int mgnns_debug_slabcopy(const void* src, void* dst, int n_rows, int pitch, int piece, int mode, int wgx,
                         mgnns_stream_t stream);

/* One thread that spins for `microseconds` of the same counter, then (slots != NULL) stamps slots[idx].  Used once per
 * process by mgnns_amd.streams to find HIP streams that sit on DIFFERENT hardware queues: a stamp on stream Y that lands
 * before the end of a spin on stream X proves X and Y do not share an in-order queue.
 */
int mgnns_debug_spin(int microseconds, uint64_t* slots, int idx, mgnns_stream_t stream);

/* ---- e: the forward's one collective -------------------------------------------------------------------------
 * The reference runs on one GPU (hard-coded cuda:0, Multi_GCN_Multihead_att.py:85,465,493; the DataParallel line at
 * engine/Multi_GCN_Multihead_Att_engine.py:365 is commented out).  Its eval forward has no cross-sample reduction, so the
 * batch shards over ranks with replicated weights and ONE all-gather of the [rows_local, num_labels] logits restores the
 * single-process result exactly.  RCCL (over xGMI inside a node) is bound at run time -- the instance the process already
 * carries (torch's) or the system librccl.so.1; MGNNS_ERR_UNSUPP if neither loads.
 *   one process per GPU:  rank 0 calls mgnns_comm_unique_id and ships the 128 bytes to the other ranks by any host channel
 *                         (a torch.distributed store, MPI, a file); every rank, with its device current, calls
 *                         mgnns_comm_init_rank (collective: returns when all ranks have joined);
 *   one process, n GPUs:  mgnns_comm_init_all(n, devices or NULL for 0..n-1, comms[n]); a single thread that issues the
 *                         all-gathers of several devices brackets them with mgnns_comm_group_start / _end.
 * mgnns_allgather_logits enqueues on `stream` (no host sync; capturable into a hipGraph); `all` is
 * [world * rows_local, num_labels] in rank order; every rank passes the same rows_local.
 */
int mgnns_comm_unique_id(void* id, size_t bytes /* >= 128 */);
int mgnns_comm_init_rank(int world, int rank, const void* id, size_t bytes, mgnns_comm_t* comm);
int mgnns_comm_init_all(int ndev, const int* devices, mgnns_comm_t* comms);
int mgnns_comm_info(mgnns_comm_t comm, int* world, int* rank);
int mgnns_allgather_logits(mgnns_comm_t comm, const float* local, int rows_local, int num_labels, float* all,
                           mgnns_stream_t stream);
int mgnns_comm_group_start(void);   /* one thread driving several communicators brackets its per-device all-gathers with these */
int mgnns_comm_group_end(void);
int mgnns_comm_destroy(mgnns_comm_t comm);

#ifdef __cplusplus
}
#endif
#endif /* MGNNS_HIP_H */
