// One image channel's label GCN as ONE persistent launch (Multi_GCN_Multihead_att.py:460-473 / 489-499 + utils/util.py:421-426):
//   adj = gen_adj(A);  X1 = LeakyReLU_0.2(adj @ (inp @ W1));  G = adj @ (X1 @ W2)     (+ the label query projection w_q(label_query))
// As separate operators this is 11-12 launches (row sums, normalisation, CSR count / scan / fill, two GEMMs, two SpMMs, the
// split-bf16 image of G, the query projection) that all run at the very start of the forward, in front of the channel's
// memory-bank kernel on the same stream: ~120 us (object) / ~200 us (scene) before the HBM-bound bank kernels could start, and
// 135 us of a 0.52-ms forward at B = 32.  Here a grid of G workgroups walks the phases with grid barriers in between:
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
// The grid must be co-resident: the launcher caps it at the CU count and the forward enqueues it first on its stream.
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
    int* counters;                           // [2], zero before the first launch; every launch leaves them zero
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

// every workgroup of the grid has finished the phase and its stores are visible: the k-th barrier waits for k * gridDim.x arrivals
__device__ __forceinline__ void lg_grid_barrier(int* counters, int k) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's write-through stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&counters[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int target = k * (int)gridDim.x;
        while (__hip_atomic_load(&counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
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
    const int NW = (int)gridDim.x * 8, gw = (int)blockIdx.x * 8 + wave;
    const int C = a.C, K0 = a.K0, N1 = a.N1, N2 = a.N2;
    const int MT = (C + 15) / 16;
    const size_t tile_bytes = SPLIT ? (size_t)2 * 16 * lg_sc(N1 > K0 ? N1 : K0) * 16 : (size_t)16 * lg_sa(N1 > K0 ? N1 : K0) * 4;
    int* s_col = reinterpret_cast<int*>(smem + tile_bytes) + wave * LG_MAXC;           // this wave's ELL row
    float* s_val = reinterpret_cast<float*>(smem + tile_bytes + 8 * LG_MAXC * sizeof(int)) + wave * LG_MAXC;

    const __amdgpu_buffer_rsrc_t r_d = lg_rsrc(a.d, (size_t)C * 4), r_nnz = lg_rsrc(a.nnz, (size_t)C * 4);
    const __amdgpu_buffer_rsrc_t r_ec = lg_rsrc(a.ell_col, (size_t)C * C * 4), r_ev = lg_rsrc(a.ell_val, (size_t)C * C * 4);
    const __amdgpu_buffer_rsrc_t r_s1 = lg_rsrc(a.S1, (size_t)C * N1 * 4), r_x1 = lg_rsrc(a.X1, (size_t)C * N1 * 4);
    const __amdgpu_buffer_rsrc_t r_s2 = lg_rsrc(a.S2, (size_t)C * N2 * 4), r_inp = lg_rsrc(a.inp, (size_t)C * K0 * 4);

    // ---- P0: degree normalisers, the first support, the label query projection --------------------------------------------------
    for (int i = gw; i < C; i += NW) {                  // d[i] = (sum_j A[i,j])^-1/2 (utils/util.py:422), one wave per row
        const float* row = a.A + (size_t)i * C;
        float s = 0.f;
        for (int j = lane; j < C; j += 64) s += row[j];
        s = wave_sum(s);
        if (lane == 0) lg_st1(r_d, i * 4, powf(s, -0.5f));
    }
    if (a.Q) {
        for (int o = gw; o < a.NLQ * a.HQ; o += NW) {   // MODEL:97 w_q(label query): one wave per output
            const int l = o / a.HQ, n = o - l * a.HQ;
            float s = 0.f;
            for (int k = lane; k < K0; k += 64) s = fmaf(a.lq[(size_t)l * K0 + k], a.wq[(size_t)n * K0 + k], s);
            s = wave_sum(s);
            if (lane == 0) a.Q[o] = s + (a.bq ? a.bq[n] : 0.f);
        }
    }
    for (int it = blockIdx.x; it < MT * (N1 / 256); it += gridDim.x)
        lg_gemm_item<SPLIT, false>(smem, r_inp, C, K0, it / (N1 / 256), it % (N1 / 256), a.w1a, a.w1b, N1, r_s1, tid, wave, lane);
    lg_grid_barrier(a.counters, 1);

    // ---- P1: normalised adjacency row i -> ELL slot, and X1[i,:] = LeakyReLU(adj[i,:] @ S1) by the same wave ----------------------
    for (int i = gw; i < C; i += NW) {
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
        __builtin_amdgcn_wave_barrier();                // the next row of this wave rewrites the LDS list
    }
    lg_grid_barrier(a.counters, 2);

    // ---- P2: S2 = X1 @ W2 ------------------------------------------------------------------------------------------------------
    for (int it = blockIdx.x; it < MT * (N2 / 256); it += gridDim.x)
        lg_gemm_item<SPLIT, true>(smem, r_x1, C, N1, it / (N2 / 256), it % (N2 / 256), a.w2a, a.w2b, N2, r_s2, tid, wave, lane);
    lg_grid_barrier(a.counters, 3);

    // ---- P3: G = adj @ S2 (+ its split-bf16 fragment-major image; rows C..16*MT of the image are zero) -----------------------------
    const int NC2 = N2 / 256, KS2 = (N2 + 31) / 32;
    for (int it = gw; it < MT * 16 * NC2; it += NW) {
        const int i = it / NC2, c0 = (it - i * NC2) * 256;
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

    // ---- re-arm the barrier counter: the last workgroup to leave (everyone is past barrier 3 by then) --------------------------------
    __syncthreads();
    if (tid == 0) {
        const int old = __hip_atomic_fetch_add(&a.counters[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (int)gridDim.x - 1) {
            __hip_atomic_store(&a.counters[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.counters[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    const size_t lds = tile + (size_t)8 * LG_MAXC * 8;
    MG_REQUIRE(lds <= 160 * 1024, "mgnns_label_gcn_fwd: %zu B of LDS needed", lds);
    // The grid barrier needs every workgroup resident at once (one 512-thread workgroup with this LDS footprint per CU), and
    // the forward runs TWO of these launches side by side (object and scene channel): two grids that each wait for more than
    // half of the CUs dead-lock each other (measured: grid 256 on both channels hangs).  Hence at most a QUARTER of the CUs
    // per launch -- 64 on an MI355X, which is also the default: the phases' work items (23 x 4, 23 x 8 tiles at C = 365)
    // divide evenly enough and the launch leaves the rest of the chip to the memory-bank kernels that start beside it.
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) {
        mgnns_set_error("mgnns_label_gcn_fwd: cannot query the CU count");
        return MGNNS_ERR_LAUNCH;
    }
    const int cap = n_cu / 4 > 0 ? n_cu / 4 : 1;
    if (grid <= 0 || grid > cap) grid = cap;
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
