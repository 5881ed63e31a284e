// One image channel's label GCN as ONE persistent launch (Multi_GCN_Multihead_att.py:460-473 / 489-499 + utils/util.py:421-426):
//   adj = gen_adj(A);  X1 = LeakyReLU_0.2(adj @ (inp @ W1));  G = adj @ (X1 @ W2)     (+ the label query projection w_q(label_query))
// As separate operators this is 11-12 launches (row sums, normalisation, CSR count / scan / fill, two GEMMs, two SpMMs, the
// split-bf16 image of G, the query projection) that all run at the very start of the forward, in front of the channel's
// memory-bank kernel on the same stream: ~120 us (object) / ~200 us (scene) before the HBM-bound bank kernels could start, and
// 135 us of a 0.52-ms forward at B = 32.  Here a grid of G workgroups pulls the work items of the four phases, in phase order,
// from ONE ticket counter:
//   P0  d = rowsum(A)^-1/2          |  S1 = inp @ W1 (16-row x 256-column MFMA work items)  |  Q = w_q(label_query)
//   P1  adj row i (normalised, non-zeros compacted in ascending column order -- no count / scan / fill passes: a row's
//       non-zeros go to an ELL slot of C entries)  and, by the same wave, X1[i,:] = LeakyReLU(sum_p val_p S1[col_p,:])
//   P2  S2 = X1 @ W2
//   P3  G[i,:] = sum_p val_p S2[col_p,:]  (+ its fragment-major split-bf16 image for the fused channel tail's read-out)
// Same association and the same per-element operation order as the separate operators (exact mode is bit-equal to them).
// SPLIT = false: exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, bit-equal to an fmaf chain) -- the fp32 parity mode;
// SPLIT = true : split-bf16 operands (hi + lo, three v_mfma_f32_16x16x32_bf16 per product, fp32-class) -- the bf16 mode.
//
// Cross-workgroup data (d, the ELL rows, S1, X1, S2) moves without fences: system-scope write-through stores, s_waitcnt
// vmcnt(0) before the barrier's arrival count (one relaxed agent-scope atomic per workgroup), cache-bypassing loads after it
// (the per-XCD L2s are not coherent with each other; an agent-scope fence per workgroup costs microseconds).
// No co-residency requirement: tickets are handed out in item order, so whoever holds an item of phase p + 1 waits (one polling
// lane, s_sleep) only for items of phase p that workgroups ALREADY RUNNING hold -- a grid of any size makes progress with any
// number of its workgroups resident, two models / two streams / a second process beside it included.  (Rounds 1-2 used a
// counting grid barrier, which hangs as soon as one workgroup of the grid cannot be scheduled.)  The wait is bounded all the
// same: after ~2^24 polls a workgroup raises the abort word (everybody drains) and the library's status word
// (mgnns_set_status_word), which the next persistent launch reports through mgnns_last_error().
#include "common.hpp"
#include "tile_bf16.hpp"
#include "tile_f32.hpp"

namespace {

constexpr int LG_THR = 512;
constexpr int LG_MAXC = 512;                 // label classes (ELL row list of a wave lives in LDS)
typedef int lg_i32x4 __attribute__((ext_vector_type(4)));

struct LgArgs {
    const float* A; int C;
    const float* inp; int K0;
    const void *w1a, *w1b; int N1;           // exact: w1a = mgnns_pack_weight_f32(W1^T [N1,K0]); split: (hi, lo) of mgnns_pack_weight_bf16_split
    const void *w2a, *w2b; int N2;
    float* G;                                // [C, N2]
    unsigned short *gp_hi, *gp_lo;           // optional: mgnns_pack_weight_bf16_split image of G ([C rows, N2 deep])
    const float *lq, *wq, *bq; int NLQ, HQ; float* Q;      // optional: Q = lq @ wq^T + bq
    float* d; int* ell_col; float* ell_val; int* nnz;      // scratch: [C], [C*C], [C*C], [C]
    float *S1, *X1, *S2;                     // scratch: [C,N1], [C,N1], [C,N2]
    int* counters;                           // 64 ints (ticket head, done[4], exited, abort: see LG_DONE ...); zero before the first launch, every launch leaves them zero
    int* status;                             // the library's status word (host-pinned, may be null)
    int rows_d, outs_q, pcs3;                // wave-granular work per queue item (multiples of 8: one or more per wave), sized to the grid
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t lg_rsrc(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00027000);
}
// COH: the buffer was written by other workgroups of THIS launch -> system-scope (cache-bypassing) access
template <bool COH> __device__ __forceinline__ f32x4 lg_ld4(__amdgpu_buffer_rsrc_t r, int off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, COH ? 17 : 0));
}
__device__ __forceinline__ float lg_ld1(__amdgpu_buffer_rsrc_t r, int off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 17));
}
__device__ __forceinline__ int lg_ld1i(__amdgpu_buffer_rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 17); }
__device__ __forceinline__ void lg_st1(__amdgpu_buffer_rsrc_t r, int off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, off, 0, 17);
}
__device__ __forceinline__ void lg_st1i(__amdgpu_buffer_rsrc_t r, int off, int v) { __builtin_amdgcn_raw_buffer_store_b32(v, r, off, 0, 17); }
__device__ __forceinline__ void lg_st4(__amdgpu_buffer_rsrc_t r, int off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lg_i32x4, v), r, off, 0, 17);
}

// counters (ints, three 64-byte lines so that ticket draws, arrival polls and the exit count do not share one):
//   [0] ticket head   [16 + p] items of phase p done   [32] workgroups that left   [33] abort
constexpr int LG_DONE = 16, LG_LEFT = 32, LG_ABORT = 33;
constexpr int LG_SPIN_LIMIT = 1 << 24;
// next work item of this workgroup (wave-uniform), -1 once the queue is empty or the launch is being aborted.  Thread 0 draws
// the ticket AFTER the next one while the current item is being worked on (`ahead`): the atomic's round trip (~1 us) hides
// behind the item.  Tickets are still handed out in item order, and a workgroup that holds two is running: the progress
// argument above is unchanged.
__device__ __forceinline__ int lg_take(int* counters, int total, int* s_ticket, int& ahead) {
    __syncthreads();                                               // the previous item is finished by every wave (LDS reuse)
    if (threadIdx.x == 0) {
        int t = ahead;                                             // (an abort is noticed at the next phase wait)
        if (t >= total) t = -1;
        *s_ticket = t;
    }
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(*s_ticket);
}
// draw the ticket after this one; called once the workgroup is about to start on its item (behind the phase hand-over, whose
// waits would otherwise include this atomic's round trip)
__device__ __forceinline__ void lg_draw_ahead(int* counters, int& ahead) {
    if (threadIdx.x == 0) ahead = __hip_atomic_fetch_add(&counters[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// all `need` items of `phase` are done and their write-through stores visible; false = aborted (bounded wait ran out)
__device__ __forceinline__ bool lg_wait_phase(int* counters, int phase, int need, int* status, int* s_ok) {
    if (threadIdx.x == 0) {
        int spins = 0, ok = 1;
        while (__hip_atomic_load(&counters[LG_DONE + phase], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            if (++spins > LG_SPIN_LIMIT ||
                ((spins & 63) == 0 && __hip_atomic_load(&counters[LG_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(&counters[LG_ABORT], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (status) __hip_atomic_store(status, MGNNS_STATUS_LABEL_GCN_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        *s_ok = ok;
    }
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(*s_ok) != 0;      // (rewritten only behind the barriers of the next lg_done / lg_take)
}
// `n` items of `phase` this workgroup has finished: their stores are acknowledged, then one relaxed arrival for all of them.
// Called when the workgroup LEAVES the phase (it draws a ticket of a later phase, or the queue is empty), not per item: waiting
// for the write-through acknowledgements after every item serialised a fabric round trip per item (measured 59 / 124 us per
// launch at C = 80 / 365 against 42 / 92 us with one wait per phase).  A workgroup that owes arrivals is running and about to
// deliver them -- before it waits for anything itself -- so the progress argument is unchanged.
__device__ __forceinline__ void lg_done(int* counters, int phase, int n) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's write-through stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&counters[LG_DONE + phase], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int lg_sa(int K) { return ((K + 15) / 16) * 16 + 4; }                  // fp32 LDS row stride: conflict-free A fragments
constexpr int lg_sc(int K) { return 4 * ((K + 31) / 32) + 2; }                   // 16-B chunk stride of the split image (== 2 mod 4)

// One work item: out[16 rows of tile mt, 256 columns of chunk nc] = src[tile rows, 0:K] @ W^T (W packed fragment-major)
template <bool SPLIT, bool COH>
__device__ __forceinline__ void lg_gemm_item(unsigned char* smem, __amdgpu_buffer_rsrc_t src, int rows, int K, int mt, int nc,
                                             const void* wa, const void* wb, int N, __amdgpu_buffer_rsrc_t dst, int tid, int wave,
                                             int lane) {
    const int r0 = mt * 16;
    const int NTt = (N + 15) / 16;
    f32x4 acc[2];
    if (!SPLIT) {
        float* As = reinterpret_cast<float*>(smem);
        const int sa = lg_sa(K), k4n = ((K + 15) / 16) * 4;
        for (int i = tid; i < 16 * k4n; i += LG_THR) {
            const int r = i / k4n, c4 = i - r * k4n;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r0 + r < rows && c4 * 4 < K) v = lg_ld4<COH>(src, ((r0 + r) * K + c4 * 4) * 4);
            *reinterpret_cast<f32x4*>(As + r * sa + c4 * 4) = v;
        }
        __syncthreads();
        mg_tile_gemm_f32<2>(acc, As, sa, K, reinterpret_cast<const float*>(wa), NTt, wave, lane, nc * 2);
    } else {
        uint4* Ah = reinterpret_cast<uint4*>(smem);
        const int sc = lg_sc(K), KS = (K + 31) / 32, c8n = 4 * KS;
        uint4* Al = Ah + 16 * sc;
        WRing<2, 3> ring;
        ring_prime(ring, KS, reinterpret_cast<const unsigned short*>(wa), reinterpret_cast<const unsigned short*>(wb), NTt, wave, lane,
                   nc * 2);                            // the weights fly through the staging of the activations
        for (int i = tid; i < 16 * c8n; i += LG_THR) {
            const int r = i / c8n, c8 = i - r * c8n;
            f32x4 u = {0.f, 0.f, 0.f, 0.f}, v = u;
            if (r0 + r < rows) {
                if (c8 * 8 < K) u = lg_ld4<COH>(src, ((r0 + r) * K + c8 * 8) * 4);
                if (c8 * 8 + 4 < K) v = lg_ld4<COH>(src, ((r0 + r) * K + c8 * 8 + 4) * 4);
            }
            unsigned short h[8], l[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = j < 4 ? u[j] : v[j - 4];
                h[j] = f2bf_t(x);
                l[j] = f2bf_t(x - bf2f_t(h[j]));
            }
            Ah[r * sc + c8] = make_uint4(h[0] | (unsigned)h[1] << 16, h[2] | (unsigned)h[3] << 16, h[4] | (unsigned)h[5] << 16, h[6] | (unsigned)h[7] << 16);
            Al[r * sc + c8] = make_uint4(l[0] | (unsigned)l[1] << 16, l[2] | (unsigned)l[3] << 16, l[4] | (unsigned)l[5] << 16, l[6] | (unsigned)l[7] << 16);
        }
        __syncthreads();
        ring_gemm(acc, ring, Ah, Al, sc, KS, reinterpret_cast<const unsigned short*>(wa), reinterpret_cast<const unsigned short*>(wb), lane);
    }
    const int crow = (lane >> 4) * 4, ccol = lane & 15;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = (wave + 8 * (nc * 2 + t)) * 16 + ccol;
        if (n < N) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r0 + crow + r < rows) lg_st1(dst, ((r0 + crow + r) * N + n) * 4, acc[t][r]);
        }
    }
    __syncthreads();                                   // the tile buffer is restaged by this workgroup's next item
}

template <bool SPLIT>
__global__ __launch_bounds__(LG_THR) void label_gcn_kernel(LgArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = a.C, K0 = a.K0, N1 = a.N1, N2 = a.N2;
    const int MT = (C + 15) / 16;
    const size_t tile_bytes = SPLIT ? (size_t)2 * 16 * lg_sc(N1 > K0 ? N1 : K0) * 16 : (size_t)16 * lg_sa(N1 > K0 ? N1 : K0) * 4;
    int* s_col = reinterpret_cast<int*>(smem + tile_bytes) + wave * LG_MAXC;           // this wave's ELL row
    float* s_val = reinterpret_cast<float*>(smem + tile_bytes + 8 * LG_MAXC * sizeof(int)) + wave * LG_MAXC;

    // buffer resources are built where a phase uses them: eight of them live across the item loop spilled ~90 SGPRs
#define LG_RSRC_D const __amdgpu_buffer_rsrc_t r_d = lg_rsrc(a.d, (size_t)C * 4)
#define LG_RSRC_ELL                                                                                          \
    const __amdgpu_buffer_rsrc_t r_nnz = lg_rsrc(a.nnz, (size_t)C * 4), r_ec = lg_rsrc(a.ell_col, (size_t)C * C * 4), \
                                 r_ev = lg_rsrc(a.ell_val, (size_t)C * C * 4)

    // ---- the item queue: phase order, the long items of a phase first ------------------------------------------------------
    //   P0  [0, nG1)           S1 = inp @ W1, 16-row x 256-column MFMA items
    //       [nG1, nG1 + nD)    degree normalisers d, rows_d rows per item (a wave per row)
    //       [.., n0)           Q = w_q(label query), outs_q outputs per item (a wave per output)
    //   P1  n1 items of 8 adjacency rows (a wave per row): normalised row -> ELL slot, X1 row
    //   P2  n2 MFMA items of S2 = X1 @ W2
    //   P3  n3 items of pcs3 (row, 256-column chunk) pieces of G = adj @ S2 (a wave per piece)
    //   (the wave-granular items are sized by the launcher to about two per workgroup and phase)
    int* s_flags = reinterpret_cast<int*>(smem + tile_bytes + (size_t)8 * LG_MAXC * 8);      // ticket, wait result (dynamic LDS: the launch may take all 160 KiB)
    int& s_ticket = s_flags[0];
    int& s_ok = s_flags[1];
    const int NC2 = N2 / 256, KS2 = (N2 + 31) / 32;
    const int RD = a.rows_d, OQ = a.outs_q, P3 = a.pcs3;
    const int nG1 = MT * (N1 / 256), nD = (C + RD - 1) / RD, nQ = a.Q ? (a.NLQ * a.HQ + OQ - 1) / OQ : 0;
    const int n0 = nG1 + nD + nQ, n1 = (C + 7) / 8, n2 = MT * NC2, n3 = (MT * 16 * NC2 + P3 - 1) / P3;
    const int total = n0 + n1 + n2 + n3;
    int phase_ok = 0;                                          // phases below this one are known to be complete
    int ahead = 0;                                             // thread 0: the ticket drawn ahead
    if (tid == 0) ahead = __hip_atomic_fetch_add(&a.counters[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int cur_phase = 0, pending = 0;                            // finished items of cur_phase not yet reported
    for (;;) {
        int it = lg_take(a.counters, total, &s_ticket, ahead);
        const int ph = it < 0 ? 4 : it < n0 ? 0 : it < n0 + n1 ? 1 : it < n0 + n1 + n2 ? 2 : 3;
        if (ph != cur_phase) {
            if (pending) lg_done(a.counters, cur_phase, pending);
            cur_phase = ph;
            pending = 0;
        }
        if (it < 0) break;
        if (ph > phase_ok) {                                   // the first item of a later phase: everything before it is complete
            if (!lg_wait_phase(a.counters, ph - 1, ph == 1 ? n0 : ph == 2 ? n1 : n2, a.status, &s_ok)) break;
            phase_ok = ph;
        }
        lg_draw_ahead(a.counters, ahead);
        if (it < n0) {
            // ---- P0 ------------------------------------------------------------------------------------------------------------
            if (it < nG1) {
                const __amdgpu_buffer_rsrc_t r_inp = lg_rsrc(a.inp, (size_t)C * K0 * 4), r_s1 = lg_rsrc(a.S1, (size_t)C * N1 * 4);
                lg_gemm_item<SPLIT, false>(smem, r_inp, C, K0, it / (N1 / 256), it % (N1 / 256), a.w1a, a.w1b, N1, r_s1, tid, wave, lane);
            } else if (it < nG1 + nD) {
                LG_RSRC_D;
                for (int i = (it - nG1) * RD + wave; i < min(C, (it - nG1 + 1) * RD); i += 8) {   // d[i] = (sum_j A[i,j])^-1/2 (utils/util.py:422)
                    const float* row = a.A + (size_t)i * C;
                    float sum = 0.f;
                    for (int j = lane; j < C; j += 64) sum += row[j];
                    sum = wave_sum(sum);
                    if (lane == 0) lg_st1(r_d, i * 4, powf(sum, -0.5f));
                }
            } else {
                const int o0 = (it - nG1 - nD) * OQ;
                for (int o = o0 + wave; o < min(a.NLQ * a.HQ, o0 + OQ); o += 8) {                 // MODEL:97 w_q(label query)
                    const int l = o / a.HQ, n = o - l * a.HQ;
                    float sum = 0.f;
                    for (int k = lane; k < K0; k += 64) sum = fmaf(a.lq[(size_t)l * K0 + k], a.wq[(size_t)n * K0 + k], sum);
                    sum = wave_sum(sum);
                    if (lane == 0) a.Q[o] = sum + (a.bq ? a.bq[n] : 0.f);
                }
            }
            ++pending;
            continue;
        }
        it -= n0;
        if (it < n1) {
            // ---- P1: normalised adjacency row i -> ELL slot, and X1[i,:] = LeakyReLU(adj[i,:] @ S1) by the same wave ----------------
            const int i = it * 8 + wave;
            if (i < C) {
                LG_RSRC_D;
                LG_RSRC_ELL;
                const __amdgpu_buffer_rsrc_t r_s1 = lg_rsrc(a.S1, (size_t)C * N1 * 4), r_x1 = lg_rsrc(a.X1, (size_t)C * N1 * 4);
                const float di = lg_ld1(r_d, i * 4);
                int cnt = 0;
                for (int j0 = 0; j0 < C; j0 += 64) {
                    const int j = j0 + lane;
                    float v = 0.f;
                    if (j < C) v = (a.A[(size_t)j * C + i] * di) * lg_ld1(r_d, j * 4);      // ((A D)^T D)[i,j], rounding order of gen_adj
                    const bool nz = j < C && v != 0.0f;
                    const unsigned long long m = __ballot(nz);
                    if (nz) {
                        const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
                        s_col[pos] = j;
                        s_val[pos] = v;
                        lg_st1i(r_ec, (i * C + pos) * 4, j);
                        lg_st1(r_ev, (i * C + pos) * 4, v);
                    }
                    cnt += __popcll(m);
                }
                if (lane == 0) lg_st1i(r_nnz, i * 4, cnt);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (int c0 = 0; c0 < N1; c0 += 256) {
                    const int off = (c0 + lane * 4) * 4;
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    int p = 0;
                    for (; p + 2 <= cnt; p += 2) {
                        const int c0_ = s_col[p], c1_ = s_col[p + 1];
                        const float w0 = s_val[p], w1 = s_val[p + 1];
                        const f32x4 x0 = lg_ld4<true>(r_s1, c0_ * N1 * 4 + off), x1 = lg_ld4<true>(r_s1, c1_ * N1 * 4 + off);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc[q] = fmaf(w0, x0[q], acc[q]);
                            acc[q] = fmaf(w1, x1[q], acc[q]);
                        }
                    }
                    if (p < cnt) {
                        const f32x4 x0 = lg_ld4<true>(r_s1, s_col[p] * N1 * 4 + off);
                        const float w0 = s_val[p];
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[q] = fmaf(w0, x0[q], acc[q]);
                    }
                    lg_st4(r_x1, i * N1 * 4 + off, f32x4{mg_act(acc[0], MGNNS_ACT_LRELU2), mg_act(acc[1], MGNNS_ACT_LRELU2),
                                                         mg_act(acc[2], MGNNS_ACT_LRELU2), mg_act(acc[3], MGNNS_ACT_LRELU2)});
                }
            }
            ++pending;
            continue;
        }
        it -= n1;
        if (it < n2) {
            // ---- P2: S2 = X1 @ W2 ----------------------------------------------------------------------------------------------
            {
                const __amdgpu_buffer_rsrc_t r_x1 = lg_rsrc(a.X1, (size_t)C * N1 * 4), r_s2 = lg_rsrc(a.S2, (size_t)C * N2 * 4);
                lg_gemm_item<SPLIT, true>(smem, r_x1, C, N1, it / NC2, it % NC2, a.w2a, a.w2b, N2, r_s2, tid, wave, lane);
            }
            ++pending;
            continue;
        }
        it -= n2;
        // ---- P3: G = adj @ S2 (+ its split-bf16 fragment-major image; rows C..16*MT of the image are zero) ---------------------------
        LG_RSRC_ELL;
        const __amdgpu_buffer_rsrc_t r_s2 = lg_rsrc(a.S2, (size_t)C * N2 * 4);
        for (int pc = it * P3 + wave; pc < min(MT * 16 * NC2, (it + 1) * P3); pc += 8) {
            const int i = pc / NC2, c0 = (pc - i * NC2) * 256;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (i < C) {
                const int cnt = lg_ld1i(r_nnz, i * 4);
                const int off = (c0 + lane * 4) * 4;
                int p = 0;
                for (; p + 2 <= cnt; p += 2) {
                    const int c0_ = lg_ld1i(r_ec, (i * C + p) * 4), c1_ = lg_ld1i(r_ec, (i * C + p + 1) * 4);
                    const float w0 = lg_ld1(r_ev, (i * C + p) * 4), w1 = lg_ld1(r_ev, (i * C + p + 1) * 4);
                    const f32x4 x0 = lg_ld4<true>(r_s2, c0_ * N2 * 4 + off), x1 = lg_ld4<true>(r_s2, c1_ * N2 * 4 + off);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[q] = fmaf(w0, x0[q], acc[q]);
                        acc[q] = fmaf(w1, x1[q], acc[q]);
                    }
                }
                if (p < cnt) {
                    const f32x4 x0 = lg_ld4<true>(r_s2, lg_ld1i(r_ec, (i * C + p) * 4) * N2 * 4 + off);
                    const float w0 = lg_ld1(r_ev, (i * C + p) * 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = fmaf(w0, x0[q], acc[q]);
                }
                *reinterpret_cast<f32x4*>(a.G + (size_t)i * N2 + c0 + lane * 4) = acc;
            }
            if (a.gp_hi) {
                const int k = c0 + lane * 4;
                const size_t e = ((((size_t)(i >> 4) * KS2 + (k >> 5)) * 64 + (i & 15) + 16 * ((k & 31) >> 3)) * 8) + (k & 7);
                unsigned short h[4], l[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    h[q] = f2bf_t(acc[q]);
                    l[q] = f2bf_t(acc[q] - bf2f_t(h[q]));
                }
                *reinterpret_cast<uint2*>(a.gp_hi + e) = make_uint2(h[0] | (unsigned)h[1] << 16, h[2] | (unsigned)h[3] << 16);
                *reinterpret_cast<uint2*>(a.gp_lo + e) = make_uint2(l[0] | (unsigned)l[1] << 16, l[2] | (unsigned)l[3] << 16);
            }
        }
        ++pending;
    }

#undef LG_RSRC_D
#undef LG_RSRC_ELL
    // ---- re-arm the queue: the last workgroup to leave (nobody holds an item or polls a counter by then) ------------------------------
    __syncthreads();
    if (tid == 0) {
        const int old = __hip_atomic_fetch_add(&a.counters[LG_LEFT], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (int)gridDim.x - 1) {
            __hip_atomic_store(&a.counters[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < 4; ++k) __hip_atomic_store(&a.counters[LG_DONE + k], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.counters[LG_LEFT], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.counters[LG_ABORT], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

extern "C" size_t mgnns_label_gcn_scratch_bytes(int C, int N1, int N2) {
    if (C <= 0 || N1 <= 0 || N2 <= 0) return 0;
    // d, nnz, ELL (col, val), S1, X1, S2, each padded to 256 B, + the two counters
    auto pad = [](size_t b) { return (b + 255) / 256 * 256; };
    return 2 * pad((size_t)C * 4) + 2 * pad((size_t)C * C * 4) + 2 * pad((size_t)C * N1 * 4) + pad((size_t)C * N2 * 4) + 256;
}

// The shape limits of mgnns_label_gcn_fwd as a predicate (the host side routes unsupported shapes to the separate operators).
extern "C" int mgnns_label_gcn_supported(int C, int K0, int N1, int N2, int split) {
    if (!(C > 0 && C <= LG_MAXC && K0 > 0 && K0 % 4 == 0 && K0 <= 1024)) return 0;
    if (!(N1 > 0 && N1 % 256 == 0 && N1 <= 1024 && N2 > 0 && N2 % 256 == 0)) return 0;
    if ((size_t)C * N2 * 4 >= ((size_t)1 << 31)) return 0;
    const int kmax = N1 > K0 ? N1 : K0;
    const size_t tile = split ? (size_t)2 * 16 * lg_sc(kmax) * 16 : (size_t)16 * lg_sa(kmax) * 4;
    return tile + (size_t)8 * LG_MAXC * 8 + 16 <= 160 * 1024 ? 1 : 0;
}

extern "C" int mgnns_label_gcn_fwd(const float* A, int C, const float* inp, int K0, int split, const void* w1a, const void* w1b,
                                   int N1, const void* w2a, const void* w2b, int N2, float* G, void* Gp_hi, void* Gp_lo,
                                   const float* label_query, int NLQ, const float* wq, const float* bq, int HQ, float* Q,
                                   void* scratch, size_t scratch_bytes, int grid, mgnns_stream_t stream) {
    MG_REQUIRE(A && inp && w1a && w2a && G && scratch, "mgnns_label_gcn_fwd: null pointer");
    MG_REQUIRE(C > 0 && C <= LG_MAXC, "mgnns_label_gcn_fwd: C=%d unsupported (1..%d)", C, LG_MAXC);
    MG_REQUIRE(K0 > 0 && K0 % 4 == 0 && K0 <= 1024, "mgnns_label_gcn_fwd: in_channel=%d must be a multiple of 4, <= 1024", K0);
    MG_REQUIRE(N1 > 0 && N1 % 256 == 0 && N1 <= 1024 && N2 > 0 && N2 % 256 == 0,
               "mgnns_label_gcn_fwd: widths %d / %d must be multiples of 256 (first <= 1024)", N1, N2);
    MG_REQUIRE(!split || (w1b && w2b), "mgnns_label_gcn_fwd: split-bf16 mode needs the lo halves of both weights");
    MG_REQUIRE((Gp_hi != nullptr) == (Gp_lo != nullptr), "mgnns_label_gcn_fwd: Gp_hi and Gp_lo go together");
    MG_REQUIRE(!Q || (label_query && wq && NLQ > 0 && HQ > 0), "mgnns_label_gcn_fwd: the query projection needs label_query, wq, NLQ, HQ");
    MG_REQUIRE(scratch_bytes >= mgnns_label_gcn_scratch_bytes(C, N1, N2) && ((uintptr_t)scratch & 255) == 0,
               "mgnns_label_gcn_fwd: scratch of %zu B (256-byte aligned) needed, %zu given", mgnns_label_gcn_scratch_bytes(C, N1, N2),
               scratch_bytes);
    MG_REQUIRE(mg_aligned16(inp) && mg_aligned16(G) && mg_aligned16(w1a) && mg_aligned16(w2a), "mgnns_label_gcn_fwd: inp / G / weights must be 16-byte aligned");
    MG_REQUIRE((size_t)C * N2 * 4 < ((size_t)1 << 31), "mgnns_label_gcn_fwd: C*N2 too large");
    auto pad = [](size_t b) { return (b + 255) / 256 * 256; };
    unsigned char* p = reinterpret_cast<unsigned char*>(scratch);
    LgArgs a;
    a.counters = reinterpret_cast<int*>(p); p += 256;
    a.d = reinterpret_cast<float*>(p); p += pad((size_t)C * 4);
    a.nnz = reinterpret_cast<int*>(p); p += pad((size_t)C * 4);
    a.ell_col = reinterpret_cast<int*>(p); p += pad((size_t)C * C * 4);
    a.ell_val = reinterpret_cast<float*>(p); p += pad((size_t)C * C * 4);
    a.S1 = reinterpret_cast<float*>(p); p += pad((size_t)C * N1 * 4);
    a.X1 = reinterpret_cast<float*>(p); p += pad((size_t)C * N1 * 4);
    a.S2 = reinterpret_cast<float*>(p);
    a.A = A; a.C = C; a.inp = inp; a.K0 = K0; a.w1a = w1a; a.w1b = w1b; a.N1 = N1; a.w2a = w2a; a.w2b = w2b; a.N2 = N2; a.G = G;
    a.gp_hi = reinterpret_cast<unsigned short*>(Gp_hi); a.gp_lo = reinterpret_cast<unsigned short*>(Gp_lo);
    a.lq = label_query; a.wq = wq; a.bq = bq; a.NLQ = NLQ; a.HQ = HQ; a.Q = Q;
    const int kmax = N1 > K0 ? N1 : K0;
    const size_t tile = split ? (size_t)2 * 16 * lg_sc(kmax) * 16 : (size_t)16 * lg_sa(kmax) * 4;
    const size_t lds = tile + (size_t)8 * LG_MAXC * 8 + 16;
    MG_REQUIRE(lds <= 160 * 1024, "mgnns_label_gcn_fwd: %zu B of LDS needed", lds);
    // Any grid is CORRECT (the item queue needs no co-residency); the default is a quarter of the CUs -- 64 on an MI355X: the
    // forward runs two of these launches side by side (object and scene channel) in front of the chip-filling memory-bank
    // kernels, a workgroup takes a whole CU (LDS), and 32 / 64 / 128 workgroups measured 0.843 / 0.848 / 0.873 ms per B = 256
    // forward, 0.533 / 0.480 / 0.492 at B = 32 (round 2, with the counting barrier).
    const int n_cu = mg_cu_count();
    if (n_cu <= 0) return MGNNS_ERR_LAUNCH;
    if (grid <= 0) grid = n_cu / 4 > 0 ? n_cu / 4 : 1;
    if (grid > 4 * n_cu) grid = 4 * n_cu;
    if (int rc = mg_check_status("mgnns_label_gcn_fwd")) return rc;     // a bounded wait of an earlier persistent launch ran out
    a.status = mg_status_word();
    // wave-granular queue items: about two per workgroup and phase (finer costs tickets, coarser leaves workgroups idle at the
    // end of a phase: 64-piece items in P3 left 18 of 64 workgroups without work at C = 365)
    auto per_item = [&](int n) { const int k = (n + 16 * grid - 1) / (16 * grid); return 8 * (k > 0 ? k : 1); };
    a.rows_d = per_item(C);
    a.outs_q = per_item(NLQ * HQ > 0 ? NLQ * HQ : 1);
    a.pcs3 = per_item(((C + 15) / 16) * 16 * (N2 / 256));
    if (split) {
        MG_DYN_LDS(label_gcn_kernel<true>, 160 * 1024);
        hipLaunchKernelGGL(label_gcn_kernel<true>, dim3(grid), dim3(LG_THR), lds, (hipStream_t)stream, a);
    } else {
        MG_DYN_LDS(label_gcn_kernel<false>, 160 * 1024);
        hipLaunchKernelGGL(label_gcn_kernel<false>, dim3(grid), dim3(LG_THR), lds, (hipStream_t)stream, a);
    }
    MG_CHECK_LAUNCH("mgnns_label_gcn_fwd");
    return 0;
}
