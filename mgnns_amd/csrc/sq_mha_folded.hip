// a8, folded variant: single-query multi-head attention WITHOUT materialising K and V.
//
// For len_q == 1 (submodules.py:55-119) the two bank projections fold algebraically into the query side:
//   s[b,h,l] = q_h . (W_k,h x_l + b_k,h) / T      = (W_k,h^T q_h) . x_l / T + const(l)      (const drops in softmax)
//   o[b,h]   = sum_l p_l (W_v,h x_l + b_v,h)      = W_v,h (sum_l p_l x_l) + b_v,h           (sum_l p_l == 1)
// so one layer is  U = qh . W_k (batched over heads),  C = softmax(U X^T / T) X,  o = C . W_v^T + b_v:
// 0.55 GFLOP instead of 61.9 GFLOP at B=256, L=196, all in exact fp32 (MFMA f32 16x16x4), bound by one read of
// the bank.  It is NOT the formulation the north-star's MFMA-utilisation metric is quoted on (sq_mha.hip / sq_mha_bf16.hip are).
// This exact-fp32 form is the folded attention of the fp32 and bf16x3 modes (model.set_attention('folded'); the default of
// bf16x3); bf16 mode runs the composed-map form of sq_mha_folded_bf16.hip by default.
//
// folded_attn_kernel: one workgroup (8 waves) per sample; wave w owns the 16-row bank tiles w, w+8, ... and reads
// each of them from memory exactly once:
//   GEMM1  S[16 rows, 16 heads] = X_tile[16, D] . U^T      (heads 8..15 are zero padding; U from LDS)
//   online softmax over rows per head (running max m, running partial sum z per lane)
//   GEMM2  Cacc[16 heads, D] += P^T[16 heads, 16 rows] . X_tile[16 rows, D]
// GEMM1 wants a lane to hold one bank ROW (K = features walked in the permuted order 16j + 4g + e, so every fetch is
// 16 contiguous bytes); GEMM2 wants a lane to hold one feature COLUMN.  The tile therefore goes through a
// wave-private LDS slab, 64 features at a time, between the two (no workgroup barrier: LDS operations of one wave
// execute in order).  P leaves GEMM1 in the C layout (col = head, rows 4g+r), which is exactly GEMM2's A operand
// for k-step r.  The eight per-wave partial results are merged through LDS with the usual exp(m_w - M) factors.
#include "common.hpp"

int mg_launch_gemm_batched(const float* X, int ldx, long sx, int M, int K, const float* W, long sw, int w_is_kn,
                           const float* bias, long sb, int N, float* Y, int ldy, long sy, int nbatch,
                           hipStream_t stream);

namespace {

constexpr int FD = 320;          // padded feature width handled (D <= 320, D % 4 == 0)
constexpr int NJ = FD / 16;      // GEMM1 k-groups of 16 features (4 MFMA k-steps each)          = 20
constexpr int NQ = FD / 64;      // GEMM2 column groups of 64 features (4 MFMA n-tiles each)     = 5
constexpr int MAXH = 8;
constexpr int WAVES = 8;
constexpr int MAXL = 208;

template <bool BF16>
__device__ __forceinline__ f32x4 load_x4(const void* bank, size_t row_off, int dim, bool ok) {
    // four consecutive features of one bank row (zero when !ok)
    if (!ok) return f32x4{0.f, 0.f, 0.f, 0.f};
    if (BF16) {
        const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(bank) + row_off + dim);
        return f32x4{__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xffff0000u),
                     __builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xffff0000u)};
    }
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(bank) + row_off + dim);
}

constexpr int USTR = FD + 4;      // U row stride in LDS (floats)
constexpr int XSTR = 68;          // slab row stride (floats): 64 features + pad
constexpr int SLAB = 16 * XSTR;   // floats per slab (16 rows x 64 features)
constexpr size_t LDS_LOOP = sizeof(float) * (MAXH * USTR + WAVES * 2 * SLAB);
constexpr size_t LDS_COMB = sizeof(float) * WAVES * MAXH * FD;
constexpr size_t LDS_MAIN = LDS_LOOP > LDS_COMB ? LDS_LOOP : LDS_COMB;
constexpr size_t LDS_BYTES = LDS_MAIN + sizeof(float) * (2 * WAVES * 16 + WAVES * MAXH + MAXH * MAXL);

template <bool BF16>
__global__ __launch_bounds__(WAVES * 64) void folded_attn_kernel(const float* __restrict__ U, const void* __restrict__ bank,
                                                                 int ld, const float* __restrict__ mask, int B, int L,
                                                                 int D, int H, float inv_temp, float* __restrict__ C,
                                                                 float* __restrict__ attn) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // loop phase: U + the wave-private transposition slabs; merge phase: comb (aliases both, after a barrier)
    float (*Us)[USTR] = reinterpret_cast<float (*)[USTR]>(smem);
    float* slabs = reinterpret_cast<float*>(smem) + MAXH * USTR;
    float (*comb)[MAXH][FD] = reinterpret_cast<float (*)[MAXH][FD]>(smem);              // per-wave sum_l p_l x_l   80 KB
    float (*mz)[WAVES][16] = reinterpret_cast<float (*)[WAVES][16]>(smem + LDS_MAIN);   // [2]: running max / sum
    float (*fac)[MAXH] = reinterpret_cast<float (*)[MAXH]>(mz + 2);                     // exp(m_w - M) / Z
    float (*s_all)[MAXL] = reinterpret_cast<float (*)[MAXL]>(fac + WAVES);              // scaled scores (attn output)

    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int dmax = BF16 ? ld : D;                // readable width of a bank row
    const size_t bank_b = (size_t)b * L * ld;
    const int ntiles = (L + 15) / 16;
    float* slab = slabs + wave * 2 * SLAB;

    auto load_tile = [&](int t, f32x4 (&xa)[NJ]) {           // lane (n, g): row 16t + n, features 16j + 4g .. + 3
        const int row = 16 * t + n;
        const bool rv = t < ntiles && row < L;
        const size_t ro = bank_b + (size_t)(rv ? row : 0) * ld;
#pragma unroll
        for (int j = 0; j < NJ; ++j) xa[j] = load_x4<BF16>(bank, ro, 16 * j + 4 * g, rv && 16 * j + 4 * g < dmax);
    };

    f32x4 xa[NJ];
    load_tile(wave, xa);
    for (int i = tid; i < MAXH * (FD / 4); i += WAVES * 64) {
        const int h = i / (FD / 4), d = (i - h * (FD / 4)) * 4;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (h < H && d < D) v = *reinterpret_cast<const f32x4*>(U + ((size_t)h * B + b) * D + d);
        *reinterpret_cast<f32x4*>(&Us[h][d]) = v;
    }
    __syncthreads();

    f32x4 acc[NQ * 4];
#pragma unroll
    for (int t = 0; t < NQ * 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, z_run = 0.f;          // head n; z_run is this lane's share (rows 4g+r) of the sum

    for (int t = wave; t < ntiles; t += WAVES) {
        const int row0 = 16 * t;
        // ---- GEMM1: S = X_tile . U^T ; A = X[row0 + n][16j + 4g + e], B = U[head n][16j + 4g + e]
        f32x4 s4[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f32x4 ub = *reinterpret_cast<const f32x4*>(&Us[n & 7][16 * j + 4 * g]);
            if (n >= MAXH) ub = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) s4[e & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[j][e], ub[e], s4[e & 1], 0, 0, 0);
        }
        // lane: head n, rows row0 + 4g + r
        float s[4];
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + 4 * g + r;
            float v = (s4[0][r] + s4[1][r]) * inv_temp;
            const bool live = row < L && (!mask || mask[(size_t)b * L + row] != 0.0f);
            v = live ? v : -INFINITY;
            s[r] = v;
            tmax = fmaxf(tmax, v);
            if (attn && n < H && row < L) s_all[n][row] = v;
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        float sc = 1.0f, p[4] = {0.f, 0.f, 0.f, 0.f};
        if (m_new != -INFINITY) {
            sc = __expf(m_run - m_new);                  // m_run == -inf -> 0
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] = __expf(s[r] - m_new);
        }
        z_run = z_run * sc + ((p[0] + p[1]) + (p[2] + p[3]));
        m_run = m_new;
        // rescale the accumulators only when some head's running max moved (wave-uniform test)
        if (__any(sc != 1.0f)) {
            float scr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) scr[r] = __shfl(sc, 4 * g + r, 64);      // head 4g+r lives in lanes n == 4g+r
#pragma unroll
            for (int tt = 0; tt < NQ * 4; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[tt][r] *= scr[r];
        }
        // ---- GEMM2: Cacc += P^T . X_tile, 64 features at a time through the slab:
        //      write lane (n, g): row n, features 64q + 16jj + 4g ..; read lane (n, g): row 4g + e, features 64q + 4n ..
        //      k-step e pairs p[e] (head n, row 4g+e) with X[row 4g+e][64q + 4n + c]
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float* sb = slab + (q & 1) * SLAB;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<f32x4*>(&sb[n * XSTR + 16 * jj + 4 * g]) = xa[4 * q + jj];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            f32x4 xb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) xb[e] = *reinterpret_cast<const f32x4*>(&sb[(4 * g + e) * XSTR + 4 * n]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[4 * q + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[e], xb[e][c], acc[4 * q + c], 0, 0, 0);
        }
        if (t + WAVES < ntiles) load_tile(t + WAVES, xa);
    }
    __syncthreads();              // every wave is done with U and its slabs: comb may overwrite them

    // ---- merge the eight waves
    // acc[4q+e][r] at lane (n, g): head 4g + r, feature 64q + 4n + e
    if (g < 2) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<f32x4*>(&comb[wave][4 * g + r][64 * q + 4 * n]) =
                    f32x4{acc[4 * q + 0][r], acc[4 * q + 1][r], acc[4 * q + 2][r], acc[4 * q + 3][r]};
    }
    {
        float z = z_run;
        z += __shfl_xor(z, 16, 64);
        z += __shfl_xor(z, 32, 64);
        if (g == 0) {
            mz[0][wave][n] = m_run;
            mz[1][wave][n] = z;
        }
    }
    __syncthreads();
    if (tid < MAXH) {
        const int h = tid;
        float M = -INFINITY;
        for (int w = 0; w < WAVES; ++w) M = fmaxf(M, mz[0][w][h]);
        float Z = 0.f, f[WAVES];
        for (int w = 0; w < WAVES; ++w) {
            f[w] = (mz[0][w][h] == -INFINITY) ? 0.f : __expf(mz[0][w][h] - M);
            Z += mz[1][w][h] * f[w];
        }
        const float iz = 1.0f / Z;                    // all rows masked: 0 * inf = NaN, as the reference's softmax
        for (int w = 0; w < WAVES; ++w) fac[w][h] = f[w] * iz;
        mz[0][0][h] = M;                              // reused below for the attn output
        mz[1][0][h] = iz;
    }
    __syncthreads();
    for (int i = tid; i < H * D; i += WAVES * 64) {
        const int h = i / D, d = i - h * D;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) v += comb[w][h][d] * fac[w][h];
        C[((size_t)h * B + b) * D + d] = v;
    }
    if (attn) {
        for (int i = tid; i < H * L; i += WAVES * 64) {
            const int h = i / L, l = i - h * L;
            const float sv = s_all[h][l];
            attn[((size_t)h * B + b) * L + l] = (sv == -INFINITY) ? 0.f * mz[1][0][h] : __expf(sv - mz[0][0][h]) * mz[1][0][h];
        }
    }
}

}  // namespace

extern "C" size_t mgnns_sq_mha_folded_workspace_bytes(int B, int D, int H) {
    return (size_t)2 * H * B * D * sizeof(float);
}

extern "C" int mgnns_sq_mha_folded_fwd(const float* qh, const void* bank, int bank_is_bf16, int ld_bank,
                                       const float* mask, int B, int L, int D, int H, int dk, const float* Wk,
                                       const float* Wv, const float* bv, void* workspace, size_t workspace_bytes,
                                       float* o, float* attn, mgnns_stream_t stream) {
    MG_REQUIRE(qh && bank && Wk && Wv && o && workspace, "mgnns_sq_mha_folded_fwd: null pointer");
    MG_REQUIRE(B >= 0 && L > 0 && L <= MAXL, "mgnns_sq_mha_folded_fwd: need 0 < L <= %d (L=%d)", MAXL, L);
    MG_REQUIRE(D > 0 && D <= FD && D % 4 == 0, "mgnns_sq_mha_folded_fwd: need D <= %d, D %% 4 == 0 (D=%d)", FD, D);
    MG_REQUIRE(H > 0 && H <= MAXH && dk > 0 && dk % 4 == 0, "mgnns_sq_mha_folded_fwd: need H <= %d, dk %% 4 == 0 (H=%d dk=%d)",
               MAXH, H, dk);
    MG_REQUIRE(bank_is_bf16 ? (ld_bank >= D && ld_bank % 4 == 0 && ld_bank <= FD) : ld_bank == D,
               "mgnns_sq_mha_folded_fwd: bad bank row stride %d (D=%d, bf16=%d)", ld_bank, D, bank_is_bf16);
    MG_REQUIRE(mg_aligned16(bank) && mg_aligned16(workspace), "mgnns_sq_mha_folded_fwd: bank/workspace must be 16-byte aligned");
    MG_REQUIRE(workspace_bytes >= mgnns_sq_mha_folded_workspace_bytes(B, D, H),
               "mgnns_sq_mha_folded_fwd: workspace too small (%zu bytes)", workspace_bytes);
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    float* Uw = reinterpret_cast<float*>(workspace);           // [H][B][D]
    float* Cw = Uw + (size_t)H * B * D;                        // [H][B][D]
    // U_h = qh[:, h*dk:(h+1)*dk] . Wk[h*dk:(h+1)*dk, :]        ([B,dk] x [dk,D], Wk slice read as a K-major [K,N] matrix)
    mg_launch_gemm_batched(qh, H * dk, dk, B, dk, Wk, (long)dk * D, 1, nullptr, 0, D, Uw, D, (long)B * D, H, s);
    MG_CHECK_LAUNCH("mgnns_sq_mha_folded_fwd(U)");
    const float inv_temp = 1.0f / sqrtf((float)dk);
    constexpr size_t LDS = LDS_BYTES;
    static_assert(LDS <= 160 * 1024, "LDS");
    MG_DYN_LDS(folded_attn_kernel<true>, LDS);
    MG_DYN_LDS(folded_attn_kernel<false>, LDS);
    if (bank_is_bf16)
        hipLaunchKernelGGL(folded_attn_kernel<true>, dim3(B), dim3(WAVES * 64), LDS, s, (const float*)Uw, bank, ld_bank, mask,
                           B, L, D, H, inv_temp, Cw, attn);
    else
        hipLaunchKernelGGL(folded_attn_kernel<false>, dim3(B), dim3(WAVES * 64), LDS, s, (const float*)Uw, bank, ld_bank, mask,
                           B, L, D, H, inv_temp, Cw, attn);
    MG_CHECK_LAUNCH("mgnns_sq_mha_folded_fwd(attn)");
    // o[:, h*dk:(h+1)*dk] = C_h . Wv[h*dk:(h+1)*dk, :]^T + bv[h*dk:(h+1)*dk]
    mg_launch_gemm_batched(Cw, D, (long)B * D, B, D, Wv, (long)dk * D, 0, bv, bv ? dk : 0, dk, o, H * dk, dk, H, s);
    MG_CHECK_LAUNCH("mgnns_sq_mha_folded_fwd(o)");
    return 0;
}
