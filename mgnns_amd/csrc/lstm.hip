// Text memory bank: embedding gather + packed 2-layer bidirectional LSTM (hidden 150), replacing
// get_text_memory_bank (Multi_GCN_Multihead_att.py:366-398: nn.Embedding, pack_padded_sequence,
// nn.LSTM, pad_packed_sequence).  PyTorch/MIOpen run this as ~1000 tiny launches per forward
// (one small GEMM + one pointwise kernel per time step per direction per layer).
//
// Here, per layer:
//   1. input projection for every VALID token of the batch at once (ragged rows packed sample-major,
//      row count on the device): Gx[r, dir*4H + n] = W_ih[dir][n,:] . x_r + b_ih[dir][n]  -- the fp32
//      MFMA GEMM with a row gather on the A side (layer 0 gathers straight from the embedding table,
//      so the embedded text never exists in HBM);
//   2. the recurrence: ONE persistent workgroup per (sample, direction) walks that sample's own
//      length.  Thread n owns gate row n of W_hh (all 150 weights in registers, loaded once), h_{t-1}
//      is broadcast from LDS, the four gates of a unit meet in LDS for the cell update.  Samples are
//      independent, so 2*B workgroups run concurrently and each stops at its own length (packed-sequence
//      semantics: the reverse direction starts at the sample's last valid token); positions >= len are
//      zero-filled by the same workgroup (pad_packed_sequence(total_length=T)).
// Gate order i, f, g, o (PyTorch).  All arithmetic fp32; the k-loop is a sequential fmaf chain.
#include "common.hpp"
#include "sq_mha_plan.hpp"

int mg_launch_linear(const float* X, int M, int K, const float* W, const float* bias, int N, float* Y, int ldy,
                     const int32_t* gather_idx, const int32_t* m_dev, hipStream_t stream);
int mg_launch_gemm_bf16(const void* A, const void* Bt, int M, int N, int Kp, const float* bias, float* C, int ldc, int act,
                        const int32_t* m_dev, hipStream_t stream, int c_bf16, void* workspace, size_t workspace_bytes);

namespace {

constexpr int HID = 150;
constexpr int G4 = 4 * HID;          // 600 gate rows
constexpr int HPAD = 152;            // h padded to a multiple of 4 for 16-B LDS broadcasts
constexpr int REC_THREADS = 768;     // 12 waves

// offs[b] = sum_{i<b} len_i (exclusive), offs[B] = total (single workgroup)
__global__ __launch_bounds__(1024) void lstm_pack_kernel(const int64_t* __restrict__ lens, int B, int T,
                                                         int32_t* __restrict__ offs, int32_t* __restrict__ order) {
    __shared__ int32_t s_off[1025];
    const int tid = threadIdx.x;
    if (tid < 8) order[B + tid] = 0;          // chain queues of the persistent recurrence (one per layer x direction)
    if (B <= 1024) {
        // common case, one sample per thread: wave scan (shuffles) + 16 wave totals, lengths in LDS as int4 for the rank count
        __shared__ __attribute__((aligned(16))) int s_l[1024];
        __shared__ int s_wave[16];
        int len = 0;
        if (tid < B) {
            long long l = lens[tid];
            len = (int)(l < 0 ? 0 : (l > T ? T : l));
        }
        s_l[tid] = tid < B ? len : -1;                         // -1: never counts as longer
        int inc = len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(inc, o, 64);
            if ((tid & 63) >= o) inc += v;
        }
        if ((tid & 63) == 63) s_wave[tid >> 6] = inc;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += s_wave[w];
        if (tid < B) offs[tid] = base + inc - len;
        if (tid == B - 1) offs[B] = base + inc;
        if (tid < B) {
            int rank = 0;
            const int4* l4 = reinterpret_cast<const int4*>(s_l);
            for (int j4 = 0; j4 < (B + 3) / 4; ++j4) {
                const int4 v = l4[j4];
                const int j = 4 * j4;
                rank += (v.x > len) || (v.x == len && j < tid);
                rank += (v.y > len) || (v.y == len && j + 1 < tid);
                rank += (v.z > len) || (v.z == len && j + 2 < tid);
                rank += (v.w > len) || (v.w == len && j + 3 < tid);
            }
            order[rank] = tid;
        }
        return;
    }
    // single workgroup: serial-chunk scan of the (<= a few thousand) lengths
    const int per = (B + 1023) / 1024;
    const int lo = tid * per, hi = min(B, lo + per);
    int s = 0;
    for (int b = lo; b < hi; ++b) {
        long long l = lens[b];
        s += (int)(l < 0 ? 0 : (l > T ? T : l));
    }
    s_off[tid + 1] = s;
    if (tid == 0) s_off[0] = 0;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = tid + 1 > o ? s_off[tid + 1 - o] : 0;
        __syncthreads();
        s_off[tid + 1] += v;
        __syncthreads();
    }
    int run = s_off[tid];
    for (int b = lo; b < hi; ++b) {
        offs[b] = run;
        long long l = lens[b];
        run += (int)(l < 0 ? 0 : (l > T ? T : l));
    }
    if (tid == 1023) offs[B] = s_off[1024];
    // order[r] = sample with the r-th longest text (ties by index): the recurrence launches 2*B workgroups on
    // 256 CUs, one per CU at a time, so the long chains must start first (rank by counting, O(B^2/1024) per thread)
    __syncthreads();
    __shared__ int s_len[4096];                                // clamped lengths (LDS: the rank loop reads every one of them per sample)
    const bool in_lds = B <= 4096;
    if (in_lds)
        for (int b = tid; b < B; b += 1024) {
            long long l = lens[b];
            s_len[b] = (int)(l < 0 ? 0 : (l > T ? T : l));
        }
    __syncthreads();
    for (int b = tid; b < B; b += 1024) {                      // clamped length of sample b = offs[b+1] - offs[b]
        const int lb = in_lds ? s_len[b] : offs[b + 1] - offs[b];
        int rank = 0;
        for (int j = 0; j < B; ++j) {
            const int lj = in_lds ? s_len[j] : offs[j + 1] - offs[j];
            rank += (lj > lb) || (lj == lb && j < b);
        }
        order[rank] = b;
    }
}

// pack_tok[r] = token id, pack_pos[r] = b*T + t for the rows r = offs[b] + t of sample b (grid = B)
__global__ __launch_bounds__(128) void lstm_fill_kernel(const int64_t* __restrict__ tok, const int64_t* __restrict__ lens,
                                                        int T, int V, const int32_t* __restrict__ offs,
                                                        int32_t* __restrict__ pack_tok, int32_t* __restrict__ pack_pos) {
    const int b = blockIdx.x;
    long long l = lens[b];
    const int len = (int)(l < 0 ? 0 : (l > T ? T : l));
    const int off = offs[b];
    for (int t = threadIdx.x; t < len; t += blockDim.x) {
        long long id = tok[(size_t)b * T + t];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);
        pack_tok[off + t] = (int32_t)id;
        pack_pos[off + t] = b * T + t;
    }
}

__device__ __forceinline__ unsigned short f2bf_rne(float x) {
    unsigned int u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// tanh(x) = 2*sigmoid(2x) - 1: one exp instead of libm's tanhf polynomial/branch mix (abs err < 2e-7)
__device__ __forceinline__ float tanhf_(float x) { return 2.0f / (1.0f + expf(-2.0f * x)) - 1.0f; }

// One workgroup (12 waves) per (sample, direction).  The recurrent GEMV gates[600] += W_hh[600,150] h[150] is split
// along K across the waves: wave w owns the 13 hidden units k in [13w, 13w+13) for ALL 600 gate rows (lane l holds
// rows l, l+64, ..: 10 x 13 = 130 weights in registers, loaded once).  Per step a wave therefore needs only ITS 13
// values of h (four 16-B LDS broadcasts instead of 38 for a row-per-thread split, which made the LDS the
// bottleneck), and leaves 600 partial sums in LDS; after one barrier the 150 cell threads add the 10 partials of
// their four gate rows in a fixed order (12 partials each), apply the cell update and publish h (LDS + the output row).
// The h rows of a chain leave through LDS in bursts of OCH steps.  A store per step sits in the wave's vector-memory queue
// between the input-projection loads of consecutive steps: memory operations retire in order and the store is conditional,
// so the compiler's wait for the NEXT step's projection row was s_waitcnt vmcnt(0) -- every step paid a store round trip
// (~1.4 us; no change to the arithmetic of a step ever moved the kernel's time).
constexpr int OCH = 32;
__device__ __forceinline__ void flush_rows(const float* s_out, int och, int s0, int s1, int len, int dir, int b, int T,
                                           float* __restrict__ out, unsigned short* __restrict__ out_bf16, int ld_bf16, int tid,
                                           int nthr) {
    for (int e = tid; e < (s1 - s0) * HID; e += nthr) {
        const int s = s0 + e / HID, j = e % HID;
        const int t = dir ? len - 1 - s : s;
        const float hh = s_out[(s % och) * HPAD + j];
        out[((size_t)b * T + t) * (2 * HID) + dir * HID + j] = hh;
        if (out_bf16) out_bf16[((size_t)b * T + t) * ld_bf16 + dir * HID + j] = f2bf_rne(hh);
    }
    mg_lds_barrier();                                                    // the rows may be overwritten by the next steps
}

constexpr int KW = 13;               // hidden units per wave (12 x 13 = 156 >= 150, the tail is zero)
constexpr int NWAVE = 12;
constexpr int RPL = 10;              // gate rows per lane (ceil(600 / 64))
constexpr int PSTR = 640;            // row stride of the partial-sum array

__global__ __launch_bounds__(REC_THREADS) void lstm_rec_kernel(const float* __restrict__ Gx, const int32_t* __restrict__ offs,
                                                               const int64_t* __restrict__ lens, int T,
                                                               const float* __restrict__ Whh_f, const float* __restrict__ Whh_b,
                                                               const float* __restrict__ bhh_f, const float* __restrict__ bhh_b,
                                                               const int32_t* __restrict__ order, float* __restrict__ out,
                                                               unsigned short* __restrict__ out_bf16, int ld_bf16) {
    __shared__ __attribute__((aligned(16))) float s_h[2][NWAVE][16];     // h, chunked per owning wave (15 + 1 pad)
    __shared__ float s_part[NWAVE][PSTR];
    __shared__ float s_act[G4];
    __shared__ float s_out[OCH][HPAD];                                   // h of the last OCH steps, flushed in bursts
    // workgroup id -> (rank, direction): both directions of the longest sample first
    const int b = order[blockIdx.x >> 1], dir = blockIdx.x & 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long l = lens[b];
    const int len = (int)(l < 0 ? 0 : (l > T ? T : l));
    const int off = offs[b];
    const float* Whh = dir ? Whh_b : Whh_f;
    const float* bhh = dir ? bhh_b : bhh_f;

    float w[RPL][KW];
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int row = lane + 64 * i;
#pragma unroll
        for (int kk = 0; kk < KW; ++kk)
            w[i][kk] = (row < G4 && wave * KW + kk < HID) ? Whh[(size_t)row * HID + wave * KW + kk] : 0.f;
    }
    // gate threads: thread n (< 600) finishes gate row n (sum of the 12 partials + input projection + bias, then the
    // row's activation: rows [300,450) are the tanh gate g, the rest sigmoid); cell threads: unit j = tid (< 150)
    const bool gate_on = tid < G4;
    const bool cell = tid < HID;
    const bool is_tanh = tid >= 2 * HID && tid < 3 * HID;
    const float bias = gate_on ? bhh[tid] : 0.f;
    if (tid < 2 * NWAVE * 16) (&s_h[0][0][0])[tid] = 0.f;
    float c = 0.f;
    __syncthreads();

    const float* gx_base = Gx + (size_t)off * (2 * G4) + dir * G4 + tid;
    float gx = 0.f;
    if (gate_on && len > 0) gx = gx_base[(size_t)(dir ? len - 1 : 0) * (2 * G4)];
    int cur = 0;
    // LDS indices kept opaque inside the loop so every access is base + IMMEDIATE offset (the compiler otherwise
    // hoists dozens of loop-invariant "base + const" addresses into registers and spills the weights)
    float* const sp = &s_part[0][0];
    int wr_idx = wave * PSTR + lane, rd_idx = tid;
    for (int s = 0; s < len; ++s) {
        asm volatile("" : "+v"(wr_idx), "+v"(rd_idx));
        // ---- 1. partial GEMV over this wave's 13 hidden units ---------------------------------------------------
        const f32x4* h4 = reinterpret_cast<const f32x4*>(s_h[cur][wave]);
        const f32x4 h0 = h4[0], h1 = h4[1], h2 = h4[2], h3 = h4[3];
        const float hv[16] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3],
                              h2[0], h2[1], h2[2], h2[3], h3[0], h3[1], h3[2], h3[3]};
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int kk = 0; kk + 1 < KW; kk += 2) {
                a0 = fmaf(w[i][kk], hv[kk], a0);
                a1 = fmaf(w[i][kk + 1], hv[kk + 1], a1);
            }
            a0 = fmaf(w[i][KW - 1], hv[KW - 1], a0);
            sp[wr_idx + 64 * i] = a0 + a1;
        }
        mg_lds_barrier();        // LDS-only barriers in the step loop: __syncthreads() implies s_waitcnt vmcnt(0), i.e. every
                                 // step would wait for the Gx prefetch and for the h stores to reach memory
        // ---- 2. every gate row: ordered sum of the 12 partials, + W_ih x + b_ih (Gx) + b_hh, activation ----------
        if (gate_on) {
            float sum = sp[rd_idx];
#pragma unroll
            for (int ww = 1; ww < NWAVE; ++ww) sum += sp[rd_idx + ww * PSTR];
            const float pre = (gx + bias) + sum;
            if (s + 1 < len) gx = gx_base[(size_t)(dir ? len - 2 - s : s + 1) * (2 * G4)];   // next step, in flight
            s_act[tid] = is_tanh ? tanhf_(pre) : sigmoidf_(pre);
        }
        mg_lds_barrier();
        // ---- 3. cell update by the 150 unit threads -------------------------------------------------------------------
        if (cell) {
            const float ig = s_act[tid], fg = s_act[HID + tid], gg = s_act[2 * HID + tid], og = s_act[3 * HID + tid];
            c = fg * c + ig * gg;
            const float hh = og * tanhf_(c);
            s_h[cur ^ 1][tid / KW][tid % KW] = hh;
            s_out[s % OCH][tid] = hh;
        }
        mg_lds_barrier();
        cur ^= 1;
        if ((s + 1) % OCH == 0 || s + 1 == len) flush_rows(&s_out[0][0], OCH, s - s % OCH, s + 1, len, dir, b, T, out, out_bf16, ld_bf16, tid, REC_THREADS);
    }
    // pad_packed_sequence(total_length=T): zeros behind the sample's length
    for (int i = tid; i < (T - len) * HID; i += REC_THREADS) {
        const int t = len + i / HID, j = i % HID;
        out[((size_t)b * T + t) * (2 * HID) + dir * HID + j] = 0.f;
        if (out_bf16) out_bf16[((size_t)b * T + t) * ld_bf16 + dir * HID + j] = 0;
    }
    // bf16 copy: the zero padding of the model dim (columns 2*HID .. ld-1) of every row, by the forward workgroup
    if (out_bf16 && dir == 0) {
        const int padw = ld_bf16 - 2 * HID;
        for (int i = tid; i < T * padw; i += REC_THREADS)
            out_bf16[((size_t)b * T + i / padw) * ld_bf16 + 2 * HID + i % padw] = 0;
    }
}

// ---- bf16-mode recurrence: the per-step GEMV on the MFMA ----------------------------------------------------------------
// Same one-workgroup-per-(sample, direction) chain as above, but gates[600] = W_hh . h runs on v_mfma_f32_4x4x4_16b_bf16
// (16 independent 4x4x4 blocks per instruction).  The gate rows are PERMUTED to n' = 4 unit + gate and wave w owns rows
// 64 w .. 64 w + 63: lane l = 4 b + j is row n' = 64 w + l = gate j of unit 16 w + b.  Its 152 weights sit in registers
// as the B operand (B_b[k][j] = W'[n'][4 ks + k], bf16); the A operand is h[4 ks .. 4 ks + 3] in every row of every block
// (one LDS broadcast read), so after 38 accumulating MFMAs every lane holds the COMPLETE pre-activation of its own row
// (four identical registers): no partial sums, no cross-wave reduction, one activation per lane; the four gates of a unit
// sit in one quad and meet through DPP, the cell state lives in a register, and the only LDS traffic and the only
// barrier of a step is the new h.  ~1.2 k cycles per step against 3.0 k of the fp32 kernel; W_hh and h are rounded to bf16
// for the product (fp32 accumulation, gates and cell state), which is what "bf16 mode" means for this kernel.
typedef short s16x4_t __attribute__((ext_vector_type(4)));
constexpr int MW = 10;                // waves: 10 x 64 = 640 >= 600 permuted gate rows
constexpr int MTHR = MW * 64;
constexpr int MKS = 38;               // k-steps of 4 (150 -> 152)
constexpr int MH = 192;               // h row in LDS (bf16), zero padded to 3 x 16 chunks of 4

__device__ __forceinline__ unsigned int pack2_bf16(float a, float b) {
    unsigned int r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#define MG_QUAD_BCAST(v, q) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (q) * 0x55, 0xF, 0xF, true))

// bf16-mode input projection, A side: Xb[r, 0:Kp] = bf16(X[gidx[r], 0:K]) (zero padded to Kp) for the r < *m_dev packed
// token rows -- the embedding / layer-0 rows gathered once, as the K-contiguous bf16 operand of the dense bf16 GEMM
constexpr int XKP = 320;              // 300 -> 5 BK slices of 64
__global__ __launch_bounds__(256) void lstm_gather_cast_kernel(const float* __restrict__ X, int K, const int32_t* __restrict__ gidx,
                                                               const int32_t* __restrict__ m_dev, int rows_max,
                                                               unsigned short* __restrict__ Xb) {
    const int per = XKP / 8;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int r = i / per, c = (i - r * per) * 8;
    int m = *m_dev;
    m = m < rows_max ? m : rows_max;
    if (r >= m) return;
    const float* src = X + (size_t)gidx[r] * K + c;
    float v[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4 q = {0.f, 0.f, 0.f, 0.f};
        if (c + 4 * h + 4 <= K) q = *reinterpret_cast<const f32x4*>(src + 4 * h);      // K % 4 == 0 (checked on the host)
        v[4 * h] = q[0]; v[4 * h + 1] = q[1]; v[4 * h + 2] = q[2]; v[4 * h + 3] = q[3];
    }
    *reinterpret_cast<uint4*>(Xb + (size_t)r * XKP + c) =
        uint4{pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])};
}

// bf16 mode, B <= 1024: pack + fill + gather as ONE launch.  lstm_pack_kernel is a single workgroup that every other
// launch of the text bank waits for (9 us alone, 16-18 us inside a replay), then lstm_fill and lstm_gather_cast follow as two
// more dependent launches -- at the head of the chain that decides the forward below 128 samples.  Here workgroup b derives ITS
// offset (sum of the lengths in front of it), ITS rank in the longest-first order and the total straight from the length
// vector (B * 8 bytes, L2 resident after the first workgroup), then writes its rows of pack_tok / pack_pos and gathers + casts
// its embedding rows (the layer-0 projection's bf16 A operand).  Same results as the three kernels, bit for bit.
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
constexpr int PREP_THR = 256;
constexpr int PREP_MAXT = 1024;
__global__ __launch_bounds__(PREP_THR) void lstm_prep_kernel(const int64_t* __restrict__ tok, const int64_t* __restrict__ lens, int B,
                                                            int T, int V, const float* __restrict__ X, int K,
                                                            int32_t* __restrict__ offs, int32_t* __restrict__ order,
                                                            int32_t* __restrict__ pack_tok, int32_t* __restrict__ pack_pos,
                                                            unsigned short* __restrict__ Xb, const float* __restrict__ plan_mask,
                                                            int32_t* __restrict__ plan) {
    __shared__ int s_len[1024];
    __shared__ int s_tok[PREP_MAXT];
    __shared__ int s_red[3][PREP_THR / 64];
    extern __shared__ int s_dyn[];                                  // the plan workgroup's scratch (mg_plan::lds_bytes(B))
    const int tid = threadIdx.x, b = blockIdx.x;
    // One workgroup MORE than samples when a packing plan is asked for (round 5): it builds the plan of the batch's text mask for the
    // packed masked attention launches (sq_mha_plan.hpp) next to the B workgroups that pack the batch -- both image->text stacks wait
    // for this stream anyway (the text bank), so neither pays a launch of its own for the plan (10-20 us each in round 4).
    if (b == B) {
        mg_plan::build<PREP_THR>(plan_mask, B, T, plan, s_dyn);
        return;
    }
    for (int i = tid; i < B; i += PREP_THR) {
        const long long l = lens[i];
        s_len[i] = (int)(l < 0 ? 0 : (l > T ? T : l));
    }
    __syncthreads();
    const int len = s_len[b];
    int off = 0, rank = 0, total = 0;
    for (int i = tid; i < B; i += PREP_THR) {
        const int li = s_len[i];
        total += li;
        off += i < b ? li : 0;
        rank += (li > len) || (li == len && i < b);
    }
    off = wave_sum_i(off); rank = wave_sum_i(rank); total = wave_sum_i(total);
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = off; s_red[1][tid >> 6] = rank; s_red[2][tid >> 6] = total; }
    for (int t = tid; t < len; t += PREP_THR) {                     // this sample's token ids (clamped like lstm_fill)
        long long id = tok[(size_t)b * T + t];
        s_tok[t] = (int)(id < 0 ? 0 : (id >= V ? V - 1 : id));
    }
    __syncthreads();
    off = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    rank = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
    total = s_red[2][0] + s_red[2][1] + s_red[2][2] + s_red[2][3];
    if (tid == 0) {
        offs[b] = off;
        order[rank] = b;
        if (b == 0) offs[B] = total;
    }
    if (b == 0 && tid < 8) order[B + tid] = 0;                      // chain queues of the persistent recurrence
    for (int t = tid; t < len; t += PREP_THR) {
        pack_tok[off + t] = s_tok[t];
        pack_pos[off + t] = b * T + t;
    }
    if (!Xb) return;                                                 // folded layer-0 projection: the recurrence reads the table by token id
    // gather + cast: (row, 8-column chunk) items, four in flight per thread
    constexpr int per = XKP / 8;
    const int items = len * per;
    for (int i0 = tid; i0 < items; i0 += 4 * PREP_THR) {
        f32x4 q[4][2];
        int rr[4], cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * PREP_THR;
            const int ic = i < items ? i : items - 1;                // clamped: the loads stay unconditional
            rr[u] = ic / per;
            cc[u] = (ic - rr[u] * per) * 8;
            const float* src = X + (size_t)s_tok[rr[u]] * K + cc[u];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                q[u][h] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (cc[u] + 4 * h + 4 <= K) q[u][h] = *reinterpret_cast<const f32x4*>(src + 4 * h);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * PREP_THR < items)
                *reinterpret_cast<uint4*>(Xb + (size_t)(off + rr[u]) * XKP + cc[u]) =
                    uint4{pack2_bf16(q[u][0][0], q[u][0][1]), pack2_bf16(q[u][0][2], q[u][0][3]), pack2_bf16(q[u][1][0], q[u][1][1]),
                          pack2_bf16(q[u][1][2], q[u][1][3])};
    }
}

// W_hh of both directions in the register layout of the kernel below: packed[((dir * MW + wave) * MKS + ks) * 64 + lane] =
// the four bf16 weights W'[64 wave + lane][4 ks .. 4 ks + 3] (gate rows permuted to n' = 4 unit + gate), so a workgroup
// fetches its 194 KB with 38 coalesced 8-byte loads per lane instead of 152 strided scalar ones.
__global__ __launch_bounds__(MTHR) void lstm_pack_whh_kernel(const float* __restrict__ Whh_f, const float* __restrict__ Whh_b,
                                                             uint2* __restrict__ packed) {
    const int dir = blockIdx.y, ks = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* Whh = dir ? Whh_b : Whh_f;
    const int unit = wave * 16 + (lane >> 2), gate = lane & 3;
    const bool row_on = unit < HID;
    const float* wrow = Whh + (size_t)(gate * HID + (row_on ? unit : 0)) * HID;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (row_on && 4 * ks + e < HID) ? wrow[4 * ks + e] : 0.f;
    packed[((size_t)(dir * MW + wave) * MKS + ks) * 64 + lane] = uint2{pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])};
}

#ifndef MG_LSTM_KM
#define MG_LSTM_KM 38                 // k-steps of the recurrence's GEMV on the matrix pipe; the other 38 - KM run as v_dot2c on the VALU (measured: loses)
#endif
#ifdef MG_LSTM_TRACE
// profiling aid (off by default): cycle sums of the step's phases, waves 0 / 9 of workgroup 0 (tools/dev/lstm_trace.py)
__device__ unsigned long long g_lstm_trace[2][8];
#define LSTM_T(i) { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); tt[i] += t1_ - t0_; t0_ = t1_; }
#else
#define LSTM_T(i)
#endif

// Persistent: workgroup w walks the (rank, direction) pairs w, 2G-1-w, 2G+w, ... of the length-sorted order (long chains
// first, each paired with a short one), the weights stay in registers across its samples.
__global__ __launch_bounds__(MTHR) void lstm_rec_bf16_kernel(const float* __restrict__ Gx, const int32_t* __restrict__ offs,
                                                             const int64_t* __restrict__ lens, int B, int T,
                                                             const uint2* __restrict__ packed,
                                                             const float* __restrict__ bhh_f, const float* __restrict__ bhh_b,
                                                             const int32_t* __restrict__ order, float* __restrict__ out,
                                                             unsigned short* __restrict__ out_bf16, int ld_bf16, int och,
                                                             unsigned short* __restrict__ next_x, int* __restrict__ queue,
                                                             const int32_t* __restrict__ tok_idx) {
    __shared__ __attribute__((aligned(16))) unsigned short s_h[2][MH];
    __shared__ int s_rank;
    extern __shared__ __attribute__((aligned(16))) float s_out[];       // [och][HPAD]: the h rows of a whole chain (och = min(T, 200))
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x, npair = 2 * B;
    const int unit = wave * 16 + (lane >> 2), gate = lane & 3;
    const bool row_on = unit < HID;
    const int row = gate * HID + (row_on ? unit : 0);              // PyTorch row (gate order i, f, g, o)
    const bool is_tanh = gate == 2;
    // The grid may be SMALLER than the number of chains (and than the chip): a recurrence workgroup is latency bound and
    // shares its CU with nothing else (its 10 waves plus an attention / memory-bank workgroup do not fit one register
    // file), so every CU it holds is a CU the chip-filling kernels of the other streams do not get.  Workgroups of even /
    // odd index serve the forward / reverse direction (weights fetched once) and pull the next chain of the length-sorted
    // order from a queue: longest first, i.e. LPT scheduling -- the makespan stays max(longest chain, steps / workgroups).
    const int dir = blockIdx.x & 1;
    s16x4_t w[MKS];
#pragma unroll
    for (int ks = 0; ks < MKS; ++ks) {
        const uint2 v = packed[((size_t)(dir * MW + wave) * MKS + ks) * 64 + lane];
        w[ks] = s16x4_t{(short)(v.x & 0xFFFFu), (short)(v.x >> 16), (short)(v.y & 0xFFFFu), (short)(v.y >> 16)};
    }
    const float bias = row_on ? (dir ? bhh_b : bhh_f)[row] : 0.f;
    (void)G; (void)npair;
    // The weights are loaded ONCE, in front of the chain loop: a real s_waitcnt here (the compiler's counter model sees the builtin)
    // tells it that none of these loads is pending at the loop header -- otherwise it guards the first use of every weight register
    // inside the step loop with vmcnt(n) down to vmcnt(0), and a vmcnt(0) in a step also waits for the input-projection rows that
    // the inline-asm loads keep three steps ahead
    __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0), expcnt / lgkmcnt untouched
    for (;;) {
        if (tid == 0) s_rank = atomicAdd(&queue[dir], 1);
        __syncthreads();
        const int rank = s_rank;
        if (rank >= B) break;
        const int b = order[rank];
        long long l = lens[b];
        const int len = (int)(l < 0 ? 0 : (l > T ? T : l));
        const int off = offs[b];
        if (tid < 2 * MH) (&s_h[0][0])[tid] = 0;
        float c = 0.f;
        __syncthreads();

        // input-projection rows of the next three steps in flight; the loads are UNCONDITIONAL (index clamped to the last
        // step; padding lanes read row 0 of their gate) so that the compiler knows how many are outstanding and waits with
        // vmcnt(2) instead of vmcnt(0) -- behind a burst of output stores a vmcnt(0) costs a store round trip
        // Row of the input projection for step st: row off + t of Gx -- or, with the layer-0 projection FOLDED into the embedding table
        // (tok_idx != nullptr: Gx is the [V, 2 * 4H] table emb . W_ih^T + b_ih, one row per vocabulary entry), row tok_idx[off + t].
        // The chain's token ids sit in a register, 64 steps per lane (one global load per 64 steps), and reach the scalar side by
        // v_readlane; the load itself takes the row as a scalar base + this lane's constant column offset.
        const int lm1 = len > 0 ? len - 1 : 0;
        const unsigned gx_col = (unsigned)((dir * G4 + row) * sizeof(float));
        int tk_blk = -1;
        int vtok = 0;
        // (through inline asm, waited for by hand: the compiler's own bookkeeping put s_waitcnt vmcnt(0) at the top of the loop --
        //  whatever the loop's shape -- i.e. a wait for the load issued one step earlier)
        auto gx_load = [&](int st, float& dst) __attribute__((always_inline)) {
            const int sc = st < lm1 ? st : lm1;
            long long r;
            if (tok_idx) {
                if ((sc >> 6) != tk_blk) {                         // (uniform, once per 64 steps)
                    tk_blk = sc >> 6;
                    const int sl = tk_blk * 64 + lane, slc = sl < lm1 ? sl : lm1;
                    // an EMPTY chain has no packed row: off == offs[b + 1] may be the total, a slot prep never wrote (a
                    // trailing zero-length sample of a padded partial batch) -- it reads table row 0, whose value no output sees
                    vtok = 0;
                    if (len > 0) vtok = tok_idx[off + (dir ? lm1 - slc : slc)];
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(vtok)::"memory");
                }
                r = __builtin_amdgcn_readlane(vtok, sc & 63);
            } else {
                r = off + (dir ? lm1 - sc : sc);
            }
            const float* p = Gx + (size_t)r * (2 * G4);            // wave-uniform
            asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(gx_col), "s"(p) : "memory");
        };
        // three registers, three steps per trip of the loop, NO rotation: a rotation (gx = gx1; gx1 = gx2; gx2 = load) moves the
        // register the newest load writes, so every step waited for the load issued one step earlier (vmcnt(0)): the
        // recurrence's step was bound by a global-load round trip instead of running three loads ahead (round 4)
#ifdef MG_LSTM_TRACE
        unsigned long long tt[4] = {0, 0, 0, 0};
#endif
        float gxa, gxb, gxc;
        gx_load(0, gxa);
        gx_load(1, gxb);
        gx_load(2, gxc);
        int cur = 0, sm = 0;                                       // sm = s % och (no integer division in the step loop)
        auto step = [&](int s, float& gx) __attribute__((always_inline)) {
#ifdef MG_LSTM_TRACE
            unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#endif
            // h as the A operand through the MFMA's A-matrix BROADCAST (cbsz = 4: all 16 blocks take their A from block abid):
            // 8-byte LDS reads put h[4 (16 c + b) .. + 3] into the four lanes of block b of register pair c, and k-step
            // ks = 16 c + b names that block -- instead of one 16-byte broadcast read (1 KB into the wave) per two k-steps
            //
            // Round 5, measured and NOT the default (MG_LSTM_KM = 38 keeps all k-steps on the matrix pipe): the k range split between
            // the matrix pipe and the VALU.  A step is the three waves of a SIMD running their 38 MFMAs back to back (3 x 38 x 8 =
            // 912 pipe cycles of ~1450); on paper v_mfma_f32_4x4x4 with one useful row in four and v_dot2c_f32_bf16 both retire
            // 32 useful MACs per cycle and SIMD and run side by side.  The last KD = 38 - KM k-steps as v_dot2c on the SAME packed
            // weight registers (a k-step's uint2 = two bf16 pairs), the h pairs WITHOUT extra data movement: lane r of every
            // 16-lane row holds pair 2 KM + 16 c + r (one 4-byte LDS read per 16 pairs) and the DPP modifier row_newbcast:n hands
            // pair n of the row to all of its lanes inside the dot instruction.  Text bank, B = 256 / 32, us: KM = 38: 175 / 168,
            // 30: 208 / 202, 26: 216 / 211, 22: 226 / 220 -- sixteen dots cost a step ~350 cycles: the dot is nowhere near one
            // issue slot on this chip (NOTES_r05 section 5).
            constexpr int KM = MG_LSTM_KM, KD = MKS - KM, NHD = (2 * KD + 15) / 16;
            static_assert(KM >= 19 && KM <= MKS, "the interleave below emits at most one dot k-step per MFMA");
            static_assert(2 * KM + 16 * NHD <= MH / 2, "the pair reads stay inside the zero-padded h row");
            const uint2* hq = reinterpret_cast<const uint2*>(s_h[cur]);
            const unsigned* hp = reinterpret_cast<const unsigned*>(s_h[cur]);
            s16x4_t hreg[3];
#pragma unroll
            for (int cc = 0; cc < (KM + 15) / 16; ++cc) {
                const uint2 hv = hq[16 * cc + (lane >> 2)];
                hreg[cc] = s16x4_t{(short)(hv.x & 0xFFFFu), (short)(hv.x >> 16), (short)(hv.y & 0xFFFFu), (short)(hv.y >> 16)};
            }
            unsigned hd[NHD > 0 ? NHD : 1];
#pragma unroll
            for (int cc = 0; cc < NHD; ++cc) hd[cc] = hp[2 * KM + 16 * cc + (lane & 15)];
#ifndef MG_LSTM_ACC
#define MG_LSTM_ACC 2
#endif
            // MG_LSTM_ACC independent accumulator chains (1 / 2 / 4 / 8 measured: 203-206 us per text bank either way -- the step is not bound by the MFMA chain)
            f32x4 a[MG_LSTM_ACC];
#pragma unroll
            for (int i = 0; i < MG_LSTM_ACC; ++i) a[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            float d0 = 0.f, d1 = 0.f;
#define MG_DOT(acc, hv, wv, n) asm("v_dot2c_f32_bf16_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(hv), "v"(wv), "n"(n))
            // MFMA k-step ks, and behind it dot k-step KM + j whenever j = ks KD / KM moves on: the two kinds alternate in the
            // instruction stream in the ratio of their counts
#define MG_K(ks)                                                                                                                    \
    if constexpr ((ks) < KM) {                                                                                                      \
        a[(ks) % MG_LSTM_ACC] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(hreg[(ks) >> 4], w[ks], a[(ks) % MG_LSTM_ACC], 4, (ks) & 15, 0); \
        if constexpr (((ks) + 1) * KD / KM > (ks) * KD / KM) {                                                                      \
            constexpr int j_ = (ks) * KD / KM;                                                                                      \
            const uint2 wv_ = __builtin_bit_cast(uint2, w[(KM + j_) < MKS ? (KM + j_) : 0]);                                        \
            MG_DOT(d0, hd[(2 * j_) >> 4], wv_.x, (2 * j_) & 15);                                                                    \
            MG_DOT(d1, hd[(2 * j_ + 1) >> 4], wv_.y, (2 * j_ + 1) & 15);                                                            \
        }                                                                                                                           \
    }
            MG_K(0) MG_K(1) MG_K(2) MG_K(3) MG_K(4) MG_K(5) MG_K(6) MG_K(7) MG_K(8) MG_K(9) MG_K(10) MG_K(11) MG_K(12) MG_K(13) MG_K(14)
            MG_K(15) MG_K(16) MG_K(17) MG_K(18) MG_K(19) MG_K(20) MG_K(21) MG_K(22) MG_K(23) MG_K(24) MG_K(25) MG_K(26) MG_K(27)
            MG_K(28) MG_K(29) MG_K(30) MG_K(31) MG_K(32) MG_K(33) MG_K(34) MG_K(35) MG_K(36) MG_K(37)
#undef MG_K
#undef MG_DOT
            static_assert(MKS == 38, "k-steps written out");
#ifdef MG_LSTM_TRACE
            asm volatile("s_nop 0" : "+v"(a[0]), "+v"(a[MG_LSTM_ACC - 1]));
            LSTM_T(0)                                              // h reads + the 38 MFMAs
#endif
            float asum = a[0][0];
#pragma unroll
            for (int i = 1; i < MG_LSTM_ACC; ++i) asum += a[i][0];
            if constexpr (KD > 0) asum += d0 + d1;
            asm volatile("s_waitcnt vmcnt(2)" : "+v"(gx)::"memory");   // the oldest of the three loads in flight (younger stores of a flush only make the wait longer)
            LSTM_T(1)                                              // wait for the input-projection row
            const float pre = (gx + bias) + asum;
            gx_load(s + 3, gx);                                    // this register's next turn is three steps away
            // one activation per lane: sigmoid(x), or tanh(x) = 2 sigmoid(2x) - 1 on the g rows
            const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(is_tanh ? -2.0f * pre : -pre));   // v_rcp_f32 (1 ulp), not a division sequence
            const float act = is_tanh ? 2.0f * sg - 1.0f : sg;
            const float ig = MG_QUAD_BCAST(act, 0), fgt = MG_QUAD_BCAST(act, 1), gg = MG_QUAD_BCAST(act, 2), og = MG_QUAD_BCAST(act, 3);
            c = fgt * c + ig * gg;
            const float hh = og * (2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * c)) - 1.0f);
            if (gate == 0 && row_on) {
                s_h[cur ^ 1][unit] = (unsigned short)(pack2_bf16(hh, 0.f) & 0xFFFFu);
                s_out[sm * HPAD + unit] = hh;
            }
            LSTM_T(2)                                              // activations, cell update, h to LDS
            mg_lds_barrier();
            LSTM_T(3)                                              // barrier
            cur ^= 1;
            const bool flush = sm + 1 == och || s + 1 == len;
            const int s0 = s - sm;
            sm = sm + 1 == och ? 0 : sm + 1;
            if (flush) {
                if (next_x) {
                    // not the last layer: the rows are only the NEXT layer's projection input -- written once, as its packed
                    // bf16 A operand ([packed row, 320], this direction's 150 columns; direction 0 also zeroes the padding)
                    for (int e = tid; e < (s + 1 - s0) * HID; e += MTHR) {
                        const int ss = s0 + e / HID, j = e % HID;
                        next_x[(size_t)(off + (dir ? len - 1 - ss : ss)) * XKP + dir * HID + j] = f2bf_rne(s_out[(ss % och) * HPAD + j]);
                    }
                    if (dir == 0)
                        for (int e = tid; e < (s + 1 - s0) * (XKP - 2 * HID); e += MTHR)
                            next_x[(size_t)(off + s0 + e / (XKP - 2 * HID)) * XKP + 2 * HID + e % (XKP - 2 * HID)] = 0;
                    mg_lds_barrier();
                } else {
                    flush_rows(s_out, och, s0, s + 1, len, dir, b, T, out, out_bf16, ld_bf16, tid, MTHR);
                }
            }
                };
        for (int s = 0; s < len; s += 3) {
            step(s, gxa);
            if (s + 1 < len) step(s + 1, gxb);
            if (s + 2 < len) step(s + 2, gxc);
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(gxa), "+v"(gxb), "+v"(gxc)::"memory");   // nothing of this chain's loads lands in a register that has moved on
#ifdef MG_LSTM_TRACE
        if (lane == 0 && (wave == 0 || wave == MW - 1) && blockIdx.x < 2 && len > (int)g_lstm_trace[wave ? 1 : 0][4]) {
            unsigned long long* g = g_lstm_trace[wave ? 1 : 0];
            for (int i = 0; i < 4; ++i) g[i] = tt[i];
            g[4] = (unsigned long long)len;
        }
#endif
        if (next_x) {
            __syncthreads();
            continue;
        }
        // pad_packed_sequence(total_length=T): zeros behind the sample's length
        for (int i = tid; i < (T - len) * HID; i += MTHR) {
            const int t = len + i / HID, j = i % HID;
            out[((size_t)b * T + t) * (2 * HID) + dir * HID + j] = 0.f;
            if (out_bf16) out_bf16[((size_t)b * T + t) * ld_bf16 + dir * HID + j] = 0;
        }
        if (out_bf16 && dir == 0) {
            const int padw = ld_bf16 - 2 * HID;
            for (int i = tid; i < T * padw; i += MTHR) out_bf16[((size_t)b * T + i / padw) * ld_bf16 + 2 * HID + i % padw] = 0;
        }
        __syncthreads();                                           // s_h / s_out are reused by the next sample
    }
}


}  // namespace

#ifdef MG_LSTM_TRACE
extern "C" int mgnns_debug_lstm_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lstm_trace), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1;
}
#endif

extern "C" size_t mgnns_bilstm_workspace_bytes(int B, int T, int hidden, int num_layers) {
    (void)num_layers;
    const size_t rows = (size_t)B * T;
    size_t bytes = rows * 8 * (size_t)hidden * sizeof(float);   // Gx [rows, 2*4*hidden]
    bytes += rows * 2 * (size_t)hidden * sizeof(float);          // layer-0 output [rows, 2*hidden]
    bytes += (2 * rows + 2 * (size_t)B + 1 + 8) * sizeof(int32_t);   // pack_tok, pack_pos, offs, order
    bytes = (bytes + 255) & ~(size_t)255;
    bytes += (size_t)2 * MW * MKS * 64 * sizeof(uint2);              // W_hh of one layer in the bf16 recurrence's register layout
    bytes = (bytes + 255) & ~(size_t)255;
    bytes += (rows + 8 * (size_t)hidden) * XKP * sizeof(unsigned short);   // bf16 mode: gathered input rows + W_ih, [*, 320] bf16
    return (bytes + 255) & ~(size_t)255;
}


// bf16 mode, weights only (cache per weight version): per layer [W_hh of both directions in the recurrence's register layout |
// W_ih of both directions as bf16 [8*hidden, 320]]
static size_t prepack_whh_bytes() { return ((size_t)2 * MW * MKS * 64 * sizeof(uint2) + 255) & ~(size_t)255; }
static size_t prepack_wih_bytes() { return ((size_t)2 * G4 * XKP * sizeof(unsigned short) + 255) & ~(size_t)255; }
extern "C" size_t mgnns_bilstm_bf16_prepack_bytes(int hidden, int num_layers) {
    (void)hidden;
    return (size_t)num_layers * (prepack_whh_bytes() + prepack_wih_bytes());
}
extern "C" int mgnns_bilstm_bf16_prepack(const float* const* w_ih_cat, const float* const* w_hh, int emb_dim, int hidden,
                                         int num_layers, void* packed, mgnns_stream_t stream) {
    MG_REQUIRE(w_ih_cat && w_hh && packed, "mgnns_bilstm_bf16_prepack: null pointer");
    MG_REQUIRE(hidden == HID && num_layers >= 1 && num_layers <= 2 && emb_dim > 0 && emb_dim <= XKP && emb_dim % 4 == 0,
               "mgnns_bilstm_bf16_prepack: unsupported hidden=%d layers=%d emb_dim=%d", hidden, num_layers, emb_dim);
    unsigned char* p = reinterpret_cast<unsigned char*>(packed);
    for (int layer = 0; layer < num_layers; ++layer) {
        MG_REQUIRE(w_ih_cat[layer] && w_hh[2 * layer] && w_hh[2 * layer + 1], "mgnns_bilstm_bf16_prepack: null weight pointer");
        hipLaunchKernelGGL(lstm_pack_whh_kernel, dim3(MKS, 2), dim3(MTHR), 0, (hipStream_t)stream, w_hh[2 * layer], w_hh[2 * layer + 1],
                           reinterpret_cast<uint2*>(p));
        p += prepack_whh_bytes();
        if (int rc = mgnns_cast_pad_bf16(w_ih_cat[layer], 2 * G4, layer == 0 ? emb_dim : 2 * HID, XKP, p, stream)) return rc;
        p += prepack_wih_bytes();
    }
    MG_CHECK_LAUNCH("mgnns_bilstm_bf16_prepack");
    return 0;
}

static int bilstm_impl(bool bf16_rec, const void* prepacked, const int64_t* tok, const int64_t* lens, int B, int T, const float* emb_table, int V,
                       int emb_dim, int hidden, int num_layers, const float* const* w_ih_cat,
                       const float* const* b_ih_cat, const float* const* w_hh, const float* const* b_hh,
                       void* workspace, size_t workspace_bytes, float* out, void* out_bf16, int ld_bf16,
                       mgnns_stream_t stream, const float* gx_table = nullptr, const float* plan_mask = nullptr,
                       int32_t* plan = nullptr) {
    MG_REQUIRE(tok && lens && emb_table && w_ih_cat && w_hh && b_ih_cat && b_hh && workspace && out,
               "mgnns_bilstm_fwd: null pointer");
    MG_REQUIRE(!out_bf16 || (ld_bf16 >= 2 * hidden && ld_bf16 % 8 == 0), "mgnns_bilstm_fwd: bad bf16 row length %d", ld_bf16);
    MG_REQUIRE(hidden == HID, "mgnns_bilstm_fwd: hidden_size=%d unsupported (150 only)", hidden);
    MG_REQUIRE(num_layers >= 1 && num_layers <= 2, "mgnns_bilstm_fwd: num_layers=%d unsupported (1..2)", num_layers);
    MG_REQUIRE(B > 0 && T > 0 && V > 0 && emb_dim > 0, "mgnns_bilstm_fwd: bad dims B=%d T=%d V=%d E=%d", B, T, V, emb_dim);
    MG_REQUIRE(workspace_bytes >= mgnns_bilstm_workspace_bytes(B, T, hidden, num_layers),
               "mgnns_bilstm_fwd: workspace too small (%zu < %zu)", workspace_bytes,
               mgnns_bilstm_workspace_bytes(B, T, hidden, num_layers));
    for (int i = 0; i < 2 * num_layers; ++i) MG_REQUIRE(w_hh[i] && b_hh[i], "mgnns_bilstm_fwd: null weight pointer %d", i);
    for (int i = 0; i < num_layers; ++i) MG_REQUIRE(w_ih_cat[i] && b_ih_cat[i], "mgnns_bilstm_fwd: null weight pointer %d", i);
    hipStream_t s = (hipStream_t)stream;
    const size_t rows = (size_t)B * T;
    float* Gx = reinterpret_cast<float*>(workspace);
    float* mid = Gx + rows * 2 * G4;
    int32_t* pack_tok = reinterpret_cast<int32_t*>(mid + rows * 2 * HID);
    int32_t* pack_pos = pack_tok + rows;
    int32_t* offs = pack_pos + rows;
    int32_t* order = offs + B + 1;
    uint2* packed = reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(workspace) +
                                             ((((unsigned char*)(order + B + 8) - (unsigned char*)workspace) + 255) & ~(size_t)255));
    unsigned short* xb = reinterpret_cast<unsigned short*>(reinterpret_cast<unsigned char*>(packed) +
                                                           (((size_t)2 * MW * MKS * 64 * sizeof(uint2) + 255) & ~(size_t)255));
    unsigned short* wb = xb + rows * XKP;
    int grid_rec = 2 * B;
    const int och = T < 200 ? T : 200;                               // h rows kept in LDS between flushes (one flush per chain for T <= 200)
    if (bf16_rec) {
        MG_DYN_LDS(lstm_rec_bf16_kernel, (size_t)och * HPAD * sizeof(float));                                                  // one persistent workgroup per CU (even count: see the kernel)
        const int n_cu = mg_cu_count();
        if (n_cu <= 0) return MGNNS_ERR_LAUNCH;
        // default: half of the CUs (LPT over the length-sorted chains keeps the makespan at ~the longest chain as long
        // as total steps / workgroups stays below it); MGNNS_LSTM_GRID overrides (bench.py decided the default)
        int cap = n_cu / 2;
        if (const int e = mg_env_int("MGNNS_LSTM_GRID", 0, 0)) cap = e;
        if (cap > n_cu) cap = n_cu;
        cap &= ~1;
        if (cap < 2) cap = 2;
        if (grid_rec > cap) grid_rec = cap;
        MG_REQUIRE(2 * num_layers <= 8, "mgnns_bilstm_bf16_fwd: num_layers=%d unsupported (<= 4)", num_layers);
    }

    // bf16 mode, up to 1024 samples of up to 1024 tokens: pack + fill + the layer-0 gather as one launch
    const bool prep_fused = bf16_rec && B <= 1024 && T <= PREP_MAXT && emb_dim % 4 == 0 && emb_dim <= XKP;
    // folded layer-0 projection (mgnns_bilstm_bf16_fold_embedding): no gather, no GEMM -- the recurrence reads table rows by token id
    const bool folded = gx_table && prep_fused;
    MG_REQUIRE(!plan == !plan_mask, "mgnns_bilstm_bf16_fwd: plan and plan_mask come together");
    MG_REQUIRE(!plan || (prep_fused && T <= mg_plan::PR),
               "mgnns_bilstm_bf16_fwd: the packing plan rides on the fused prep launch (bf16 recurrence, B <= 1024, T <= %d, emb_dim %% 4 == 0); "
               "build it with mgnns_sq_mha32_plan instead", mg_plan::PR);
    if (prep_fused) {
        const size_t plan_lds = plan ? mg_plan::lds_bytes(B) : 0;
        if (plan) MG_DYN_LDS(lstm_prep_kernel, plan_lds);
        hipLaunchKernelGGL(lstm_prep_kernel, dim3(B + (plan ? 1 : 0)), dim3(PREP_THR), plan_lds, s, tok, lens, B, T, V, emb_table, emb_dim, offs,
                           order, pack_tok, pack_pos, folded ? (unsigned short*)nullptr : xb, plan_mask, plan);
    } else {
        hipLaunchKernelGGL(lstm_pack_kernel, dim3(1), dim3(1024), 0, s, lens, B, T, offs, order);
        hipLaunchKernelGGL(lstm_fill_kernel, dim3(B), dim3(128), 0, s, tok, lens, T, V, (const int32_t*)offs, pack_tok, pack_pos);
    }
    for (int layer = 0; layer < num_layers; ++layer) {
        const float* X = layer == 0 ? emb_table : mid;
        const int K = layer == 0 ? emb_dim : 2 * HID;
        const int32_t* gidx = layer == 0 ? pack_tok : pack_pos;
        float* dst = (layer == num_layers - 1) ? out : mid;
        // both directions' input projections in one GEMM: W_ih = [forward ; reverse] stacked to [2*4H, in]
        const unsigned char* pp = prepacked ? reinterpret_cast<const unsigned char*>(prepacked) + (size_t)layer * (prepack_whh_bytes() + prepack_wih_bytes())
                                            : nullptr;
        if (folded && layer == 0) {
            // nothing to compute: Gx of this layer is the table
        } else if (bf16_rec && K % 4 == 0 && K <= XKP) {
            // bf16 mode: the projection on the dense bf16 GEMM (bf16 operands, fp32 accumulation and output): 8 us instead of 42.
            // Layer 0 gathers + casts the embedding rows; the later layers find their operand written by the recurrence below.
            if (layer == 0 && !prep_fused)
                hipLaunchKernelGGL(lstm_gather_cast_kernel, dim3((unsigned)((rows * (XKP / 8) + 255) / 256)), dim3(256), 0, s, X, K, gidx,
                                   (const int32_t*)(offs + B), (int)rows, xb);
            const void* wih = wb;
            if (pp) wih = pp + prepack_whh_bytes();
            else if (int rc = mgnns_cast_pad_bf16(w_ih_cat[layer], 2 * G4, K, XKP, wb, stream)) return rc;
            if (int rc = mg_launch_gemm_bf16(xb, wih, (int)rows, 2 * G4, XKP, b_ih_cat[layer], Gx, 2 * G4, MGNNS_ACT_NONE, offs + B, s, 0, nullptr, 0)) return rc;
        } else {
            mg_launch_linear(X, (int)rows, K, w_ih_cat[layer], b_ih_cat[layer], 2 * G4, Gx, 2 * G4, gidx, offs + B, s);
        }
        unsigned short* obf = (layer == num_layers - 1) ? reinterpret_cast<unsigned short*>(out_bf16) : (unsigned short*)nullptr;
        if (bf16_rec) {
            const uint2* whh = packed;
            if (pp) whh = reinterpret_cast<const uint2*>(pp);
            else hipLaunchKernelGGL(lstm_pack_whh_kernel, dim3(MKS, 2), dim3(MTHR), 0, s, w_hh[2 * layer], w_hh[2 * layer + 1], packed);
            const bool tab = folded && layer == 0;
            hipLaunchKernelGGL(lstm_rec_bf16_kernel, dim3(grid_rec), dim3(MTHR), (size_t)och * HPAD * sizeof(float), s,
                               tab ? gx_table : (const float*)Gx, (const int32_t*)offs, lens, B, T, whh, b_hh[2 * layer], b_hh[2 * layer + 1],
                               (const int32_t*)order, dst, obf, ld_bf16, och,
                               (layer + 1 < num_layers && 2 * HID <= XKP) ? xb : (unsigned short*)nullptr, order + B + 2 * layer,
                               tab ? (const int32_t*)pack_tok : (const int32_t*)nullptr);
        }
        else
            hipLaunchKernelGGL(lstm_rec_kernel, dim3(2 * B), dim3(REC_THREADS), 0, s, (const float*)Gx, (const int32_t*)offs, lens,
                               T, w_hh[2 * layer], w_hh[2 * layer + 1], b_hh[2 * layer], b_hh[2 * layer + 1],
                               (const int32_t*)order, dst, obf, ld_bf16);
    }
    MG_CHECK_LAUNCH("mgnns_bilstm_fwd");
    return 0;
}

extern "C" int mgnns_bilstm_fwd(const int64_t* tok, const int64_t* lens, int B, int T, const float* emb_table, int V,
                                int emb_dim, int hidden, int num_layers, const float* const* w_ih_cat,
                                const float* const* b_ih_cat, const float* const* w_hh, const float* const* b_hh,
                                void* workspace, size_t workspace_bytes, float* out, void* out_bf16, int ld_bf16,
                                mgnns_stream_t stream) {
    return bilstm_impl(false, nullptr, tok, lens, B, T, emb_table, V, emb_dim, hidden, num_layers, w_ih_cat, b_ih_cat, w_hh, b_hh, workspace,
                       workspace_bytes, out, out_bf16, ld_bf16, stream);
}

extern "C" int mgnns_bilstm_bf16_fwd(const int64_t* tok, const int64_t* lens, int B, int T, const float* emb_table, int V,
                                     int emb_dim, int hidden, int num_layers, const float* const* w_ih_cat,
                                     const float* const* b_ih_cat, const float* const* w_hh, const float* const* b_hh,
                                     void* workspace, size_t workspace_bytes, float* out, void* out_bf16, int ld_bf16,
                                     const void* prepacked, const float* plan_mask, int32_t* plan, mgnns_stream_t stream) {
    return bilstm_impl(true, prepacked, tok, lens, B, T, emb_table, V, emb_dim, hidden, num_layers, w_ih_cat, b_ih_cat, w_hh, b_hh, workspace,
                       workspace_bytes, out, out_bf16, ld_bf16, stream, nullptr, plan_mask, plan);
}

// ---- the layer-0 input projection folded into the embedding table (weights only: once per weight version) -------------------
// Gx0[row] = bf16(emb[tok[row]]) . bf16(W_ih0)^T + b_ih0 depends on the token id alone: table[v] = bf16(emb[v]) . bf16(W_ih0)^T + b_ih0
// for every vocabulary entry ([V, 2 * 4H] fp32: 97 MB for V = 20 154) is computed ONCE by the same GEMM (same k order per element:
// the rows of the forward come out bit for bit), and a forward reads its rows by token id: no gather / cast of embedding rows, no
// GEMM in front of the first recurrence -- 15-19 us off the head of the chain the forward follows.
extern "C" size_t mgnns_bilstm_bf16_table_bytes(int V, int hidden) { return (size_t)(V > 0 ? V : 0) * 8 * (size_t)hidden * sizeof(float); }
extern "C" size_t mgnns_bilstm_bf16_fold_workspace_bytes(int V) {
    return (((size_t)(V > 0 ? V : 0) + 8 * (size_t)HID) * XKP * sizeof(unsigned short) + 255) & ~(size_t)255;
}
extern "C" int mgnns_bilstm_bf16_fold_embedding(const float* emb_table, int V, int emb_dim, int hidden, const float* w_ih_cat0,
                                                const float* b_ih_cat0, void* workspace, size_t workspace_bytes, float* table,
                                                mgnns_stream_t stream) {
    MG_REQUIRE(emb_table && w_ih_cat0 && b_ih_cat0 && workspace && table, "mgnns_bilstm_bf16_fold_embedding: null pointer");
    MG_REQUIRE(hidden == HID && V > 0 && emb_dim > 0 && emb_dim <= XKP && emb_dim % 4 == 0,
               "mgnns_bilstm_bf16_fold_embedding: unsupported hidden=%d V=%d emb_dim=%d", hidden, V, emb_dim);
    MG_REQUIRE(workspace_bytes >= mgnns_bilstm_bf16_fold_workspace_bytes(V), "mgnns_bilstm_bf16_fold_embedding: workspace too small (%zu < %zu)",
               workspace_bytes, mgnns_bilstm_bf16_fold_workspace_bytes(V));
    unsigned short* eb = reinterpret_cast<unsigned short*>(workspace);
    unsigned short* wb = eb + (size_t)V * XKP;
    if (int rc = mgnns_cast_pad_bf16(emb_table, V, emb_dim, XKP, eb, stream)) return rc;
    if (int rc = mgnns_cast_pad_bf16(w_ih_cat0, 2 * G4, emb_dim, XKP, wb, stream)) return rc;
    return mg_launch_gemm_bf16(eb, wb, V, 2 * G4, XKP, b_ih_cat0, table, 2 * G4, MGNNS_ACT_NONE, nullptr, (hipStream_t)stream, 0, nullptr, 0);
}
extern "C" int mgnns_bilstm_bf16_table_fwd(const int64_t* tok, const int64_t* lens, int B, int T, const float* emb_table, int V,
                                           int emb_dim, int hidden, int num_layers, const float* const* w_ih_cat,
                                           const float* const* b_ih_cat, const float* const* w_hh, const float* const* b_hh,
                                           void* workspace, size_t workspace_bytes, float* out, void* out_bf16, int ld_bf16,
                                           const void* prepacked, const float* gx_table, const float* plan_mask, int32_t* plan,
                                           mgnns_stream_t stream) {
    MG_REQUIRE(gx_table, "mgnns_bilstm_bf16_table_fwd: null table (mgnns_bilstm_bf16_fold_embedding makes it)");
    return bilstm_impl(true, prepacked, tok, lens, B, T, emb_table, V, emb_dim, hidden, num_layers, w_ih_cat, b_ih_cat, w_hh, b_hh, workspace,
                       workspace_bytes, out, out_bf16, ld_bf16, stream, gx_table, plan_mask, plan);
}
