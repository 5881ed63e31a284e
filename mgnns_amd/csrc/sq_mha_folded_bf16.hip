// a8, folded variant on the bf16 matrix pipe: single-query multi-head attention with the K / V projections AND the
// query / output projections composed into the maps either side of it (submodules.py:55-119, len_q == 1).
//
//   s[b,h,l] = q_h . (W_k,h x_l + b_k,h) / T   with   q_h = W_q,h x + b_q,h
//            = u_h . x_l / T + const(l)            u_h = (W_k,h^T W_q,h) x + W_k,h^T b_q,h          (const drops in the softmax)
//   fc(o)    = sum_h fc_h (W_v,h c_h + b_v,h) + b  c_h = sum_l p[h,l] x_l                            (sum_l p == 1)
//            = sum_h (fc_h W_v,h) c_h + (fc b_v + b)
// The composed maps M = [W_k,h^T W_q,h]_h (H*D x D) and N = [fc_h W_v,h]_h (D x H*D) are built once per weight version on
// the host side (fusion.py) and run inside the layer tails (mha_tail.hip), which leaves THIS kernel with
//   S = X U^T (L x D x H),  P = softmax(S / T + mask),  C = P X (H x L x D)
// per sample: 0.5 MFLOP-class, one read of the sample's memory bank, instead of the 242 MFLOP of projecting K and V.
// It is not the formulation the north-star's MFMA-utilisation figure is quoted on (sq_mha_bf16.hip is); it computes the same
// attention (tests: the reference's goldens).
//
// One 512-thread workgroup per sample.  The bank [L <= 208, 320] bf16 goes ONCE into LDS by LDS-DMA (672-B row stride:
// conflict-free for the A-fragment reads AND for the transposing reads below).  GEMM 1: row tiles over the waves, A = X tile,
// B = U (8 heads, zero padded to 16) -> scores [row, head].  Softmax: one wave per head; the probabilities go to an LDS
// image in GEMM 2's A-fragment order.  GEMM 2: C[head, feature] = P X with the contraction over ROWS, i.e. X is wanted
// feature-major: ds_read_b64_tr_b16 (gfx950's transposing LDS read) returns, per 16-lane group, the 4 x 16 block of four
// rows as one column of four per lane.  The k-slot -> row assignment of a k-step is ours to choose (P is laid out to match):
// slot (g, j) = row 32 ks + 16 (j >> 2) + 4 g + (j & 3), which keeps the four rows of a read adjacent and the two lane
// groups of a 32-lane pass on disjoint banks.
#include "common.hpp"
#include "tile_bf16.hpp"

#ifdef MG_FOLD_TRACE
// profiling aid (off by default; tools/dev/fold_trace.py): s_memtime stamps of wave 0 of workgroups 0 and 129 at the phase boundaries
__device__ unsigned long long g_fold_trace[2][16];
#define MG_FSTAMP(i) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 129)) g_fold_trace[blockIdx.x ? 1 : 0][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MG_FSTAMP(i) do { } while (0)
#endif

namespace {

constexpr int MT = 13;                  // row tiles of 16 (L <= 208)
constexpr int LMAX = MT * 16;
constexpr int KP = 320;                 // model dim padded to 10 k-steps of 32
constexpr int KSTEPS = KP / 32;
constexpr int CH = KP / 8;              // 40 16-byte chunks per bank row
constexpr int LSTR = 42;                // LDS row stride in chunks
constexpr int ROWB = LSTR * 16;         // 672 B
constexpr int MAXH = 8;
constexpr int NTHR = 512;
constexpr int NT2 = KP / 16;            // feature tiles of GEMM 2
constexpr int KS2 = 7;                  // k-steps (32 rows) of GEMM 2
constexpr int PROW = KS2 * 32;          // P image row (bf16)

constexpr size_t OFF_U = (size_t)LMAX * ROWB;                              // bf16 [MAXH][336]
constexpr size_t OFF_SC = OFF_U + MAXH * ROWB;                             // float [MAXH][LMAX]
constexpr size_t OFF_P = OFF_SC + MAXH * LMAX * sizeof(float);             // bf16 [16][PROW]
constexpr size_t OFF_MB = OFF_P + 16 * PROW * 2;                           // float [LMAX] mask bias: 0 or -inf
constexpr size_t OFF_INT = OFF_MB + LMAX * sizeof(float);
constexpr size_t SMEM_BYTES = OFF_INT + 64;
static_assert(SMEM_BYTES <= 160 * 1024, "LDS");
static_assert(OFF_U % 16 == 0 && OFF_SC % 16 == 0 && OFF_P % 16 == 0, "LDS alignment");

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// The transposing read (TR): lane i of a 16-lane group passes the address of row (i >> 2), features 4 (i & 3) .. + 3 of the
// group's 4-row x 16-feature block and receives feature i of rows 0 .. 3.  TR = false reads the same values with 2-byte loads
// (MGNNS_FOLD_TR=0: the cross-check of the transposing form in the tests).
template <bool TR>
__global__ __launch_bounds__(NTHR) void folded_attn_bf16_kernel(const float* __restrict__ U, const unsigned short* __restrict__ bank,
                                                                const float* __restrict__ mask, int B, int L, int D, int H,
                                                                float inv_temp, unsigned short* __restrict__ C, int ldc,
                                                                float* __restrict__ attn) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    int* s_int = reinterpret_cast<int*>(smem + OFF_INT);
    float* s_sc = reinterpret_cast<float*>(smem + OFF_SC);
    unsigned short* s_p = reinterpret_cast<unsigned short*>(smem + OFF_P);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int b = blockIdx.x;
    const uint4* xb = reinterpret_cast<const uint4*>(bank) + (size_t)b * L * CH;
    MG_FSTAMP(0);

    // ---- live rows (row tiles behind the last unmasked position are never staged: their probability is exactly 0) and the
    //      mask as a score bias, one global round trip
    float* s_mb = reinterpret_cast<float*>(smem + OFF_MB);
    int lvalid = L;
    if (mask) {
        if (tid == 0) s_int[0] = 0;
        __syncthreads();
        if (tid < LMAX) {
            const bool live = tid < L && mask[(size_t)b * L + tid] != 0.0f;
            s_mb[tid] = live ? 0.0f : -INFINITY;
            const unsigned long long bal = __ballot(live);
            if (lane == 0 && bal) atomicMax(s_int, 64 * wave + 64 - __builtin_clzll(bal));
        }
        __syncthreads();
        lvalid = s_int[0];
    } else if (tid < LMAX) {
        s_mb[tid] = tid < L ? 0.0f : -INFINITY;              // (published by the staging barrier below)
    }
    const int n_mt = lvalid > 0 ? (lvalid + 15) >> 4 : 1;
    const int rows_live = n_mt * 16;
    MG_FSTAMP(1);

    // ---- the bank: LDS-DMA, 1 KiB of contiguous LDS per instruction from per-lane addresses; lanes on the two pad chunks of
    //      a row are off, rows >= L read the zero padding at the end of bank row 0
    {
        const int total = rows_live * LSTR;
        for (int pc = wave; pc * 64 < total; pc += NTHR / 64) {
            const int gi = pc * 64 + lane;
            const int row = gi / LSTR, c = gi - row * LSTR;
            if (gi < total && c < CH) {
                const uint4* src = row < L ? xb + (size_t)row * CH + c : xb + (CH - 1);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(uintptr_t)(smem + (size_t)pc * 1024),
                                                 16, 0, 0);
            }
        }
    }
    MG_FSTAMP(2);
    // ---- the composed query rows: fp32 [H][D] -> bf16 [H][320] (zero padded), P image zeroed
    for (int i = tid; i < H * CH; i += NTHR) {
        const int h = i / CH, c = i - h * CH;
        const float* src = U + (size_t)b * H * D + (size_t)h * D + 8 * c;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 8 * c + j < D ? src[j] : 0.f;
        uint4 q;
        q.x = f2bf2_t(v[0], v[1]); q.y = f2bf2_t(v[2], v[3]); q.z = f2bf2_t(v[4], v[5]); q.w = f2bf2_t(v[6], v[7]);
        *reinterpret_cast<uint4*>(smem + OFF_U + (size_t)h * ROWB + (size_t)c * 16) = q;
    }
    for (int i = tid; i < 16 * PROW / 8; i += NTHR) reinterpret_cast<uint4*>(s_p)[i] = make_uint4(0u, 0u, 0u, 0u);
    MG_FSTAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    MG_FSTAMP(4);

    // ---- GEMM 1: scores[row, head] = X U^T ------------------------------------------------------------------------------
    {
        bf16x8 ub[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            uint4 q = *reinterpret_cast<const uint4*>(smem + OFF_U + (size_t)(n & (MAXH - 1)) * ROWB + (size_t)(4 * ks + g) * 16);
            if (n >= H) q = make_uint4(0u, 0u, 0u, 0u);
            ub[ks] = __builtin_bit_cast(bf16x8, q);
        }
        for (int t = wave; t < n_mt; t += NTHR / 64) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const unsigned char* xr = smem + (size_t)(16 * t + n) * ROWB + (size_t)g * 16;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks += 2) {
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xr + ks * 64));
                const bf16x8 a1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xr + ks * 64 + 64));
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, ub[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, ub[ks + 1], acc1, 0, 0, 0);
            }
            if (n < H) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * t + 4 * g + r;
                    s_sc[n * LMAX + row] = (acc0[r] + acc1[r]) * inv_temp + s_mb[row];
                }
            }
        }
    }
    __syncthreads();
    MG_FSTAMP(5);

    // ---- softmax over the rows, one wave per head; probabilities -> GEMM 2's A image (+ the attn output) --------------
    for (int h = wave; h < H; h += NTHR / 64) {
        float v[4];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int l = lane + 64 * j;
            v[j] = l < rows_live ? s_sc[h * LMAX + l] : -INFINITY;
            m = fmaxf(m, v[j]);
        }
        m = wave_max_dpp(m);
        float e[4], z = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            e[j] = __expf(v[j] - m);               // every row masked: -inf - -inf = NaN, as the reference's softmax
            z += e[j];
        }
        z = wave_sum_dpp(z);
        const float iz = 1.0f / z;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int l = lane + 64 * j;
            const float p = e[j] * iz;
            if (attn && l < L) attn[((size_t)h * B + b) * L + l] = p;
            if (l < PROW) {
                const int ks = l >> 5, r32 = l & 31;
                s_p[h * PROW + ((ks * 4 + ((r32 >> 2) & 3)) << 3) + ((r32 >> 4) << 2) + (r32 & 3)] = f2bf_t(p);
            }
        }
    }
    __syncthreads();
    MG_FSTAMP(6);

    // ---- GEMM 2: C[head, feature] = P X; feature tiles over the waves (0-3: three, 4-7: two) --------------------------------
    {
        const int nks = (n_mt + 1) >> 1;
        f32x4 acc[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        // Branch-free: a wave's missing third tile re-reads its first one (result dropped), rows behind the staged ones are
        // clamped to the last staged row (their probabilities are exactly zero in the P image; staged data is finite), so that
        // every read of a k-step is in flight before the first MFMA waits.
        int ft[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) ft[t] = 16 * (wave + 8 * t < NT2 ? wave + 8 * t : wave);
        const int last_row = rows_live - 1;
        s16x4 lo[3], hi[3];
        bf16x8 a;
        auto fetch = [&](int ks) {
            a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(s_p + n * PROW + ((ks * 4 + g) << 3)));
            const int r_lo = 32 * ks + 4 * g, r_hi = r_lo + 16;
            int rl, rh;
            if (TR) {
                rl = min(r_lo + (n >> 2), last_row);
                rh = min(r_hi + (n >> 2), last_row);
            } else {
                rl = r_lo;
                rh = r_hi;
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (TR) {
                    const unsigned char* pl = smem + (size_t)rl * ROWB + (size_t)(ft[t] + 4 * (n & 3)) * 2;
                    const unsigned char* ph = smem + (size_t)rh * ROWB + (size_t)(ft[t] + 4 * (n & 3)) * 2;
                    lo[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)pl);
                    hi[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)ph);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        lo[t][k] = *reinterpret_cast<const short*>(smem + (size_t)min(rl + k, last_row) * ROWB + (size_t)(ft[t] + n) * 2);
                        hi[t][k] = *reinterpret_cast<const short*>(smem + (size_t)min(rh + k, last_row) * ROWB + (size_t)(ft[t] + n) * 2);
                    }
                }
            }
        };
        fetch(0);
        for (int ks = 0; ks < nks; ++ks) {
            const bf16x8 ac = a;
            s16x8 bv[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) bv[t] = s16x8{lo[t][0], lo[t][1], lo[t][2], lo[t][3], hi[t][0], hi[t][1], hi[t][2], hi[t][3]};
            if (ks + 1 < nks) fetch(ks + 1);
#pragma unroll
            for (int t = 0; t < 3; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac, __builtin_bit_cast(bf16x8, bv[t]), acc[t], 0, 0, 0);
        }
        MG_FSTAMP(7);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int nt = wave + 8 * t;
            const int f = 16 * nt + n;
            if (nt < NT2 && f < D) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int h = 4 * g + r;
                    if (h < H) C[(size_t)b * ldc + h * D + f] = f2bf_t(acc[t][r]);
                }
            }
        }
        for (int i = H * D + tid; i < ldc; i += NTHR) C[(size_t)b * ldc + i] = 0;      // k padding of the tail's first product
        MG_FSTAMP(8);
    }
}

}  // namespace

#ifdef MG_FOLD_TRACE
extern "C" int mgnns_debug_fold_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fold_trace), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : 1;
}
#endif

// U: fp32 [B, H*D] (head h at h*D: the composed query rows, bias included); bank: bf16 [B, L, 320] (zero padded);
// mask: [B, L] fp32 (0 = masked) or null; C: bf16 [B, ldc], ldc >= H*D, ldc % 8 == 0 (probability-weighted bank rows per head at
// h*D, zeros behind H*D);
// attn: fp32 [H*B, L] (the reference's layout, MODEL: submodules.py:76-82) or null.  inv_temp = 1 / sqrt(d_k).
extern "C" int mgnns_sq_mha_folded_bf16_fwd(const float* U, const void* bank_bf16, const float* mask, int B, int L, int D, int H,
                                            float inv_temp, void* C_bf16, int ldc, float* attn, mgnns_stream_t stream) {
    MG_REQUIRE(U && bank_bf16 && C_bf16, "mgnns_sq_mha_folded_bf16_fwd: null pointer");
    MG_REQUIRE(B >= 0 && L > 0 && L <= LMAX, "mgnns_sq_mha_folded_bf16_fwd: need 0 < L <= %d (L=%d)", LMAX, L);
    MG_REQUIRE(D > 0 && D <= KP && D % 4 == 0, "mgnns_sq_mha_folded_bf16_fwd: need D <= %d, D %% 4 == 0 (D=%d)", KP, D);
    MG_REQUIRE(H > 0 && H <= MAXH, "mgnns_sq_mha_folded_bf16_fwd: need 0 < H <= %d (H=%d)", MAXH, H);
    MG_REQUIRE(mg_aligned16(bank_bf16), "mgnns_sq_mha_folded_bf16_fwd: the bank must be 16-byte aligned");
    MG_REQUIRE(ldc >= H * D && ldc % 8 == 0, "mgnns_sq_mha_folded_bf16_fwd: ldc=%d (>= H*D = %d, multiple of 8)", ldc, H * D);
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool tr = mg_env_int("MGNNS_FOLD_TR", 1, 6) != 0;
    MG_DYN_LDS(folded_attn_bf16_kernel<true>, SMEM_BYTES);
    MG_DYN_LDS(folded_attn_bf16_kernel<false>, SMEM_BYTES);
    if (tr)
        hipLaunchKernelGGL(folded_attn_bf16_kernel<true>, dim3(B), dim3(NTHR), SMEM_BYTES, s, U,
                           static_cast<const unsigned short*>(bank_bf16), mask, B, L, D, H, inv_temp,
                           static_cast<unsigned short*>(C_bf16), ldc, attn);
    else
        hipLaunchKernelGGL(folded_attn_bf16_kernel<false>, dim3(B), dim3(NTHR), SMEM_BYTES, s, U,
                           static_cast<const unsigned short*>(bank_bf16), mask, B, L, D, H, inv_temp,
                           static_cast<unsigned short*>(C_bf16), ldc, attn);
    MG_CHECK_LAUNCH("mgnns_sq_mha_folded_bf16_fwd");
    return 0;
}
