// Row f4: the CNN trunks either side of the path -- torchvision-style ResNet-101 (objects) and ResNet-50
// (Places365) cut after layer4 (MODEL:274-294, 586-595, 629-630), eval mode, as four kernels:
//
//   conv_fold_bn_kernel     one-off weight preparation: BatchNorm (running statistics) folded into the convolution,
//                           w'[o, (kh, kw, c)] = bf16(w[o, c, kh, kw] * gamma[o] / sqrt(var[o] + eps)), b'[o] = beta - mean * scale
//   stem_conv7_kernel       conv1 7x7 / stride 2 / pad 3 (3 -> 64) + bn1 + ReLU, NCHW fp32 image -> NHWC bf16
//   maxpool3x3s2_kernel     MaxPool2d(3, 2, 1) on NHWC bf16
//   conv_igemm_ws_kernel    every 1x1 / 3x3 convolution of the bottlenecks as an implicit GEMM on
//                           v_mfma_f32_16x16x32_bf16:  Y[m, o] = relu?( sum_k X[pixel(m, tap(k)), c(k)] * w'[o, k] + b'[o] + R[m, o] )
//
// Activations are NHWC bf16 ([B, H, W, C], C contiguous = the GEMM's K axis), accumulation fp32.  The implicit GEMM is
// the dense bf16 GEMM of gemm_bf16.hip (256 x 128 workgroup tile, 8 compute + 4 producer waves, BK = 64, three-stage LDS-DMA ring, XOR chunk
// swizzle, XCD-aware super tiles) with the A rows resolved per BK slice: C_in is a power of two >= 64, so one BK slice
// lies inside one filter tap (kh, kw) and an A row of the slice is 128 contiguous bytes of the input pixel
// (oh*s - p + kh, ow*s - p + kw), or 16 zero bytes from g_zero16 when that pixel is padding.  No im2col buffer exists.
// Narrow layers (C_out = 64) use a 256 x 64 tile (NJ = 2).  The last convolution of a trunk writes the
// [B, 2048, 14, 14] fp32 NCHW map the fusion path reads (out_nchw = 1).
#include "common.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ unsigned int f2bf(float f) {       // round to nearest even; inputs are finite
    unsigned int u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf2f(unsigned int h) { return __uint_as_float(h << 16); }
// two fp32 -> packed bf16x2 (round to nearest even) in ONE instruction (no builtin for it on gfx950); the shift / add form
// above costs five VALU operations per value, which made the convolution epilogue VALU-bound
__device__ __forceinline__ unsigned int pack2(float a, float b) {
    unsigned int r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_fold_bn_kernel(const float* __restrict__ w, const float* __restrict__ cbias,
                                                           int Cout, int Cin, int KH, int KW, int ld,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean, const float* __restrict__ var,
                                                           float eps, int nchw_k, unsigned short* __restrict__ wt,
                                                           float* __restrict__ bias) {
    const int o = blockIdx.x;
    const float scale = gamma ? gamma[o] / sqrtf(var[o] + eps) : 1.0f;
    const int K = KH * KW * Cin;
    for (int k = threadIdx.x; k < ld; k += blockDim.x) {
        float v = 0.f;
        if (k < K) {
            int c, kh, kw;
            if (nchw_k) { c = k / (KH * KW); kh = (k / KW) % KH; kw = k % KW; }      // k = (c, kh, kw): the stem
            else { kh = k / (KW * Cin); kw = (k / Cin) % KW; c = k % Cin; }          // k = (kh, kw, c): NHWC implicit GEMM
            v = w[(((size_t)o * Cin + c) * KH + kh) * KW + kw] * scale;
        }
        wt[(size_t)o * ld + k] = (unsigned short)f2bf(v);
    }
    if (threadIdx.x == 0) {
        const float cb = cbias ? cbias[o] : 0.f;
        bias[o] = gamma ? (beta[o] + (cb - mean[o]) * scale) : cb;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stem: one workgroup = 8 output rows x 32 output columns x 64 channels of one image; the 21 x 69 x 3 input patch is
// staged in LDS as bf16; a wave owns two output rows (four 16-pixel MFMA tiles).  K = (c, kh, kw) = 147, padded to 160:
// the A fragment of a lane (8 consecutive k) is gathered with 16-bit LDS reads through per-lane patch offsets.
constexpr int ST_ROWS = 8, ST_COLS = 32, ST_PR = 2 * ST_ROWS + 5, ST_PC = 2 * ST_COLS + 5, ST_K = 160;

__global__ __launch_bounds__(256) void stem_conv7_kernel(const float* __restrict__ img, int H, int W, int OH, int OW,
                                                         const unsigned short* __restrict__ wt, const float* __restrict__ bias,
                                                         unsigned short* __restrict__ y) {
    __shared__ unsigned short patch[3 * ST_PR * ST_PC + 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, or0 = blockIdx.y * ST_ROWS, oc0 = blockIdx.x * ST_COLS;
    const int ir0 = 2 * or0 - 3, ic0 = 2 * oc0 - 3;
    const float* src = img + (size_t)b * 3 * H * W;
    // all loads of the patch in flight before the first conversion (a rolled loop pays one HBM round trip per element)
    constexpr int NE = 3 * ST_PR * ST_PC, NIT = (NE + 255) / 256;
    float pv[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int e = tid + 256 * i;
        const int c = e / (ST_PR * ST_PC), rem = e - c * (ST_PR * ST_PC);
        const int r = rem / ST_PC, cc = rem - r * ST_PC;
        const int ir = ir0 + r, ic = ic0 + cc;
        pv[i] = 0.f;
        if (e < NE && (unsigned)ir < (unsigned)H && (unsigned)ic < (unsigned)W) pv[i] = src[((size_t)c * H + ir) * W + ic];
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int e = tid + 256 * i;
        if (e < NE) patch[e] = (unsigned short)pack2(pv[i], 0.f);
    }
    const int fr = lane & 15, fg = lane >> 4;
    // weights: lane (n = fr, k-group fg) of column tile jt, k-step s
    uint4 bw[4][5];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int s = 0; s < 5; ++s)
            bw[jt][s] = *reinterpret_cast<const uint4*>(wt + (size_t)(jt * 16 + fr) * ST_K + s * 32 + fg * 8);
    int koff[5][8];
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int jx = 0; jx < 8; ++jx) {
            const int k = s * 32 + fg * 8 + jx;
            const int c = k / 49, kh = (k % 49) / 7, kw = k % 7;
            koff[s][jx] = k < 147 ? (c * ST_PR + kh) * ST_PC + kw : 0;
        }
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
        const int orl = 2 * wave + (t >> 1), ocl = (t & 1) * 16 + fr;
        const int pbase = (2 * orl) * ST_PC + 2 * ocl;
        f32x4 acc[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) acc[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            uint4 av;
            unsigned int e[8];
#pragma unroll
            for (int jx = 0; jx < 8; ++jx) e[jx] = patch[pbase + koff[s][jx]];
            av.x = e[0] | (e[1] << 16);
            av.y = e[2] | (e[3] << 16);
            av.z = e[4] | (e[5] << 16);
            av.w = e[6] | (e[7] << 16);
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
                acc[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bw[jt][s]),
                                                                  __builtin_bit_cast(bf16x8, av), acc[jt], 0, 0, 0);
        }
        // acc[jt][r] = out[pixel fr of the tile][channel jt*16 + 4 fg + r]
        const int orow = or0 + orl, ocol = oc0 + ocl;
        if (orow < OH && ocol < OW) {
            unsigned short* dst = y + (((size_t)b * OH + orow) * OW + ocol) * 64;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                const int n = jt * 16 + fg * 4;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
                uint2 o;
                o.x = pack2(fmaxf(acc[jt][0] + bv[0], 0.f), fmaxf(acc[jt][1] + bv[1], 0.f));
                o.y = pack2(fmaxf(acc[jt][2] + bv[2], 0.f), fmaxf(acc[jt][3] + bv[3], 0.f));
                *reinterpret_cast<uint2*>(dst + n) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// MaxPool2d(kernel 3, stride 2, padding 1), NHWC bf16: one thread = one output pixel x 8 channels (16 bytes)
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const unsigned short* __restrict__ x, int B, int H, int W, int C,
                                                           int OH, int OW, unsigned short* __restrict__ y) {
    const int cg = C >> 3;
    const size_t total = (size_t)B * OH * OW * cg;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        size_t p = i / cg;
        const int ow = (int)(p % OW);
        p /= OW;
        const int oh = (int)(p % OH), b = (int)(p / OH);
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int ih = 2 * oh - 1 + dh;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int iw = 2 * ow - 1 + dw;
                if ((unsigned)iw >= (unsigned)W) continue;
                const uint4 v = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + ih) * W + iw) * C + g * 8);
                const unsigned int u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    m[2 * j] = fmaxf(m[2 * j], bf2f(u[j] & 0xFFFFu));
                    m[2 * j + 1] = fmaxf(m[2 * j + 1], bf2f(u[j] >> 16));
                }
            }
        }
        uint4 o;
        o.x = (__float_as_uint(m[0]) >> 16) | (__float_as_uint(m[1]) & 0xFFFF0000u);
        o.y = (__float_as_uint(m[2]) >> 16) | (__float_as_uint(m[3]) & 0xFFFF0000u);
        o.z = (__float_as_uint(m[4]) >> 16) | (__float_as_uint(m[5]) & 0xFFFF0000u);
        o.w = (__float_as_uint(m[6]) >> 16) | (__float_as_uint(m[7]) & 0xFFFF0000u);
        *reinterpret_cast<uint4*>(y + (((size_t)b * OH + oh) * OW + ow) * C + g * 8) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution: persistent and warp specialised.  One workgroup per CU walks its tiles (XCD-aware order: the 32
// workgroups of an XCD work on one super tile of row blocks x all column tiles at a time) and treats the BK slices of ALL its
// tiles as one stream through the three-stage LDS-DMA ring, so the slices of the next tile are requested while the current
// tile still computes and the epilogue's loads / stores overlap the next tile's first slices.  Twelve waves, three per SIMD:
// waves 8..11 (one per SIMD) are PRODUCERS -- they only issue the LDS-DMA pieces of the slice stream (a quarter of every
// slice each; pieces past the end of the stream fetch a zero source so the operation count behind a slice is fixed and
// s_waitcnt vmcnt(N) is exact) and wait for them; waves 0..7 only read fragments, issue MFMAs and run the epilogue.  One
// s_barrier per slice publishes slice g+1 and frees the stage of slice g.
// History (all measured): one tile per workgroup with the DMA issue inside the eight compute waves: 6.96 ms per
// 32 ResNet-101 images; 16-byte epilogue 6.12; persistent stream + inline-asm LDS reads 5.19 (a compute wave still stalled
// ~1.2-1.4 k cycles per slice in the issue of its six DMA pieces -- the vector-memory path was backed up -- against ~0.7 k
// cycles of its own MFMA work, in-kernel timer); v_cvt_pk_bf16_f32 epilogue 4.96; producer waves 4.73.
constexpr int TM = 256, BK = 64, NSTAGE = 3;
constexpr int A_BYTES = TM * BK * 2, A_PIECES = A_BYTES / 1024;      // 32 KB = 32 DMA pieces of 8 rows x 128 B

typedef int i32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const unsigned short* x;      // [B, H, W, Cin] bf16
    const unsigned short* wt;     // [Cout, KH*KW*Cin] bf16, k = (kh, kw, c)
    const float* bias;            // [Cout]
    const unsigned short* res;    // [M, Cout] bf16 or null
    void* y;                      // [M, Cout] bf16, or [B, Cout, OH*OW] fp32 when out_nchw
    int H, W, Cin, cin_shift, OH, OW, Cout, KH, KW, stride, pad, M, relu;
    float inv_ohw, inv_ow;
    unsigned int ybytes;
    int nrb, nct, rps, jmax;      // tile map: row blocks, column tiles, row blocks per super tile, virtual tiles per XCD
};

constexpr int NTHR_WS = 768, NPROD = 4;

template <int NJ, bool HAS_RES, bool OUT_NCHW>
__global__ __launch_bounds__(NTHR_WS) void conv_igemm_ws_kernel(const ConvArgs a) {
    constexpr int TN = 32 * NJ, B_BYTES = TN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int B_PIECES = B_BYTES / 1024;
    constexpr int APP = A_PIECES / NPROD, BPP = B_PIECES / NPROD, PPP = APP + BPP;   // pieces per producer and slice: 8 + 4 (2)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, jj0 = blockIdx.x >> 3, jstep = gridDim.x >> 3;
    const int per = a.rps * a.nct;
    const int K = a.KH * a.KW * a.Cin, nk = K / BK;
    const int ohw = a.OH * a.OW;

    auto decode = [&](int j, int& m0, int& n0) {
        const int sl = j / per, within = j - sl * per;
        const int rb = (sl * 8 + xcd) * a.rps + within / a.nct, ct = within % a.nct;
        m0 = rb * TM;
        n0 = ct * TN;
        return j < a.jmax && rb < a.nrb;
    };
    auto next_valid = [&](int j, int& m0, int& n0) {
        while (j < a.jmax && !decode(j, m0, n0)) j += jstep;
        return j;
    };
    int ntiles = 0;
    {
        int m0, n0;
        for (int j = jj0; j < a.jmax; j += jstep) ntiles += decode(j, m0, n0) ? 1 : 0;
    }
    if (ntiles == 0) return;
    const int S = ntiles * nk;

    if (wave >= 8) {
        // =============================== producer ===============================
        const int q = wave - 8;
        const int row_in = lane >> 3, slot = lane & 7;
        const int chunk = slot ^ (4 * (q & 1) + (row_in >> 1));    // pieces q + 4 i: p & 1 == q & 1
        int im0 = 0, in0 = 0, ikt = 0, ig = 0;
        int ij = next_valid(jj0, im0, in0);
        int ih0[APP], iw0[APP];
        long long base[APP];
        auto rows_of_tile = [&]() {
            const int b0 = im0 / ohw, rem0 = im0 - b0 * ohw;
#pragma unroll
            for (int i = 0; i < APP; ++i) {
                const int off = (q + NPROD * i) * 8 + row_in;
                if (im0 + off < a.M) {
                    int x = rem0 + off;
                    int kb = (int)((float)x * a.inv_ohw);
                    int r = x - kb * ohw;
                    if (r < 0) { r += ohw; --kb; } else if (r >= ohw) { r -= ohw; ++kb; }
                    int oh = (int)((float)r * a.inv_ow);
                    int ow = r - oh * a.OW;
                    if (ow < 0) { ow += a.OW; --oh; } else if (ow >= a.OW) { ow -= a.OW; ++oh; }
                    ih0[i] = oh * a.stride - a.pad;
                    iw0[i] = ow * a.stride - a.pad;
                    base[i] = (((long long)(b0 + kb) * a.H + ih0[i]) * a.W + iw0[i]) * a.Cin;
                } else {
                    ih0[i] = -(1 << 20);
                    iw0[i] = 0;
                    base[i] = 0;
                }
            }
        };
        rows_of_tile();
        const unsigned short* zsrc = reinterpret_cast<const unsigned short*>(g_zero16);
        auto issue = [&]() {
            unsigned char* sb = smem + (size_t)(ig % NSTAGE) * STAGE_BYTES;
            const bool live = ig < S;
            const int k0 = ikt * BK;
            const int tap = k0 >> a.cin_shift, c0 = k0 & (a.Cin - 1);
            const int kh = a.KW == 3 ? (tap * 11) >> 5 : 0, kw = tap - kh * a.KW;
            const int toff = (kh * a.W + kw) * a.Cin + c0;
            auto dma = [&](const unsigned short* src, int p) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(uintptr_t)(sb + (size_t)p * 1024), 16, 0, 0);
            };
#pragma unroll
            for (int i = 0; i < APP; ++i) {
                const bool ok = live && (unsigned)(ih0[i] + kh) < (unsigned)a.H && (unsigned)(iw0[i] + kw) < (unsigned)a.W;
                dma(ok ? a.x + (base[i] + toff + chunk * 8) : zsrc, q + NPROD * i);
            }
#pragma unroll
            for (int i = 0; i < BPP; ++i) {
                int row = in0 + (q + NPROD * i) * 8 + row_in;
                row = row < a.Cout ? row : a.Cout - 1;
                dma(live ? a.wt + (size_t)row * K + k0 + chunk * 8 : zsrc, A_PIECES + q + NPROD * i);
            }
            ++ig;
            if (++ikt == nk && ig < S) {
                ikt = 0;
                ij = next_valid(ij + jstep, im0, in0);
                rows_of_tile();
            }
        };
        issue();
        issue();
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PPP) : "memory");     // slice 0 landed
        issue();
        for (int g = 0; g < S; ++g) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PPP) : "memory"); // slice g+1 landed; stage of slice g is free
            issue();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // dummy pieces must not outlive the workgroup's LDS
        return;
    }

    // =============================== compute ===============================
    const int wr = wave >> 1, wc = wave & 1;
    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    const unsigned lds0 = mg_lds_addr(smem);
    unsigned aoff[2], boff[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        aoff[s2] = lds0 + ((wr * 64 + fr) * 8 + ((4 * s2 + fg) ^ ((fr >> 1) & 7))) * 16;
        boff[s2] = lds0 + A_BYTES + ((wc * 16 * NJ + fr) * 8 + ((4 * s2 + fg) ^ ((fr >> 1) & 7))) * 16;
    }
    // with a residual the 64 x 64 bf16 residual tile is prefetched a slice ahead (32 registers): the fragments are then
    // single buffered (read k-step, MFMAs, read k-step, MFMAs) so that the wave stays inside the 168 registers three waves
    // per SIMD leave it; these layers (1x1, K <= 512, + residual) are bound by their activation traffic, not by the MFMAs
    constexpr int NBUF = HAS_RES ? 1 : 2;
    u32x4 av[NBUF][4], bv[NBUF][NJ];
    auto reads = [&](int stage, int s2, int buf) {
        const unsigned so = (unsigned)stage * STAGE_BYTES;
        av[buf][0] = mg_lds_read128<0>(aoff[s2] + so);
        av[buf][1] = mg_lds_read128<2048>(aoff[s2] + so);
        av[buf][2] = mg_lds_read128<4096>(aoff[s2] + so);
        av[buf][3] = mg_lds_read128<6144>(aoff[s2] + so);
        bv[buf][0] = mg_lds_read128<0>(boff[s2] + so);
        bv[buf][1] = mg_lds_read128<2048>(boff[s2] + so);
        if constexpr (NJ == 4) {
            bv[buf][2] = mg_lds_read128<4096>(boff[s2] + so);
            bv[buf][3] = mg_lds_read128<6144>(boff[s2] + so);
        }
    };
    auto mmas = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
                acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bv[buf][jj]),
                                                                    __builtin_bit_cast(bf16x8, av[buf][i]), acc[i][jj], 0, 0, 0);
    };
    int cm0 = 0, cn0 = 0, ckt = 0;
    int cj = next_valid(jj0, cm0, cn0);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.ybytes, 0x00027000);
    const int ncl = wc * 16 * NJ + (fg & 1) * 16 + (fg >> 1) * 8;
    u32x4 rv[NJ / 2][4];
    f32x4 bs[NJ / 2][2];
    // ordinary loads (the compiler places their waits): no DMA sits in a compute wave's vector-memory queue, and an
    // inline-asm load is only safe while nothing spills -- under register pressure hipcc reuses the destination
    // registers of a load it cannot see is still in flight
    auto gload = [&](const void* ptr) { return *reinterpret_cast<const u32x4*>(ptr); };

    asm volatile("s_barrier" ::: "memory");                        // slice 0 landed (the producers waited for it)
    if constexpr (NBUF == 2) reads(0, 0, 0);
    for (int g = 0; g < S; ++g) {
        const bool last = ckt == nk - 1;
        if (last) {                                                // bias / residual of the tile this slice closes, a slice ahead of their use
#pragma unroll
            for (int jp = 0; jp < NJ / 2; ++jp) {
                int n = cn0 + ncl + jp * 32;
                n = n < a.Cout ? n : a.Cout - 8;
                bs[jp][0] = __builtin_bit_cast(f32x4, gload(a.bias + n));
                bs[jp][1] = __builtin_bit_cast(f32x4, gload(a.bias + n + 4));
                if (HAS_RES) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int m = cm0 + wr * 64 + i * 16 + fr;
                        m = m < a.M ? m : a.M - 1;
                        rv[jp][i] = gload(a.res + (size_t)m * a.Cout + n);
                    }
                }
            }
        }
        if constexpr (NBUF == 2) {
            reads(g % NSTAGE, 1, 1);
            mg_lds_wait<4 + NJ>();
            __builtin_amdgcn_sched_barrier(0);
            mmas(0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // this wave is done with the stage of slice g; slice g+1 landed
            reads((g + 1) % NSTAGE, 0, 0);
            mg_lds_wait<4 + NJ>();
            __builtin_amdgcn_sched_barrier(0);
            mmas(1);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            reads(g % NSTAGE, 0, 0);
            mg_lds_wait<0>();
            __builtin_amdgcn_sched_barrier(0);
            mmas(0);
            __builtin_amdgcn_sched_barrier(0);
            reads(g % NSTAGE, 1, 0);
            mg_lds_wait<0>();
            __builtin_amdgcn_sched_barrier(0);
            mmas(0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_barrier" ::: "memory");                // done with the stage of slice g; slice g+1 landed
        }
        if (++ckt < nk) continue;
        ckt = 0;
#pragma unroll
        for (int jp = 0; jp < NJ / 2; ++jp) {
            const int n = cn0 + ncl + jp * 32;
            const bool ncok = n < a.Cout;
            const f32x4 b0 = bs[jp][0], b1 = bs[jp][1];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 lo = acc[i][2 * jp], hi = acc[i][2 * jp + 1];
                asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\t"
                             "v_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7\n\ts_nop 1"
                             : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
                acc[i][2 * jp] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc[i][2 * jp + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int m = cm0 + wr * 64 + i * 16 + fr;
                const bool ok = ncok && m < a.M;
                float o[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[r] = lo[r] + b0[r];
                    o[4 + r] = hi[r] + b1[r];
                }
                if (HAS_RES) {
                    const unsigned int u[4] = {rv[jp][i][0], rv[jp][i][1], rv[jp][i][2], rv[jp][i][3]};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o[2 * r] += __uint_as_float(u[r] << 16);
                        o[2 * r + 1] += __uint_as_float(u[r] & 0xFFFF0000u);
                    }
                }
                if (a.relu) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) o[r] = fmaxf(o[r], 0.f);
                }
                if (OUT_NCHW) {
                    const int mm = ok ? m : 0;
                    const int b = mm / ohw, p = mm - b * ohw;
                    const unsigned int off = ok ? (unsigned int)((((size_t)b * a.Cout + n) * ohw + p) * 4) : 0xFFFFFFF0u;
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, o[r]), y_rsrc, ok ? off + (unsigned)(r * ohw * 4) : off, 0, 0);
                } else {
                    i32x4 ov;
                    ov[0] = (int)pack2(o[0], o[1]);
                    ov[1] = (int)pack2(o[2], o[3]);
                    ov[2] = (int)pack2(o[4], o[5]);
                    ov[3] = (int)pack2(o[6], o[7]);
                    const unsigned int off = ok ? (unsigned int)(((size_t)m * a.Cout + n) * 2) : 0xFFFFFFF0u;
                    __builtin_amdgcn_raw_buffer_store_b128(ov, y_rsrc, off, 0, 0);
                }
            }
        }
        cj = next_valid(cj + jstep, cm0, cn0);
    }
}

template <int NJ, bool HAS_RES, bool OUT_NCHW>
int launch_conv3(ConvArgs& a, hipStream_t stream, int n_cu) {
    constexpr int TN = 32 * NJ;
    constexpr size_t SMEM = (size_t)NSTAGE * (A_BYTES + TN * BK * 2);
    auto kern = conv_igemm_ws_kernel<NJ, HAS_RES, OUT_NCHW>;
    MG_DYN_LDS(kern, SMEM);
    a.nrb = (a.M + TM - 1) / TM;
    a.nct = (a.Cout + TN - 1) / TN;
    int rps = 32 / a.nct;
    if (rps < 1) rps = 1;
    if (rps > a.nrb) rps = a.nrb;
    a.rps = rps;
    const int supers = (a.nrb + rps - 1) / rps;
    a.jmax = ((supers + 7) / 8) * rps * a.nct;                    // virtual tiles per XCD
    int per_xcd = n_cu / 8;                                        // one persistent workgroup per CU
    if (per_xcd < 1) per_xcd = 1;
    if (per_xcd > a.jmax) per_xcd = a.jmax;
    hipLaunchKernelGGL(kern, dim3(8 * per_xcd), dim3(NTHR_WS), SMEM, stream, a);
    return 0;
}

template <int NJ>
int launch_conv(ConvArgs& a, bool out_nchw, hipStream_t stream, int n_cu) {
    if (out_nchw) return a.res ? launch_conv3<NJ, true, true>(a, stream, n_cu) : launch_conv3<NJ, false, true>(a, stream, n_cu);
    return a.res ? launch_conv3<NJ, true, false>(a, stream, n_cu) : launch_conv3<NJ, false, false>(a, stream, n_cu);
}

}  // namespace

extern "C" int mgnns_conv_fold_bn_bf16(const float* w, const float* conv_bias, int Cout, int Cin, int KH, int KW,
                                       const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                                       int k_order, int ld, void* wt, float* bias, mgnns_stream_t stream) {
    MG_REQUIRE(w && wt && bias, "mgnns_conv_fold_bn_bf16: null pointer");
    MG_REQUIRE(!gamma || (beta && mean && var), "mgnns_conv_fold_bn_bf16: gamma given without beta / mean / var");
    MG_REQUIRE(Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && ld >= KH * KW * Cin && (k_order == 0 || k_order == 1),
               "mgnns_conv_fold_bn_bf16: bad dims Cout=%d Cin=%d KH=%d KW=%d ld=%d k_order=%d", Cout, Cin, KH, KW, ld, k_order);
    hipLaunchKernelGGL(conv_fold_bn_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, w, conv_bias, Cout, Cin, KH, KW, ld,
                       gamma, beta, mean, var, eps, k_order, reinterpret_cast<unsigned short*>(wt), bias);
    MG_CHECK_LAUNCH("mgnns_conv_fold_bn_bf16");
    return 0;
}

extern "C" int mgnns_stem_conv7_fwd(const float* img, int B, int H, int W, const void* wt, const float* bias, void* y,
                                    mgnns_stream_t stream) {
    MG_REQUIRE(img && wt && bias && y, "mgnns_stem_conv7_fwd: null pointer");
    MG_REQUIRE(B >= 0 && H >= 7 && W >= 7 && B <= 65535, "mgnns_stem_conv7_fwd: bad dims B=%d H=%d W=%d", B, H, W);
    MG_REQUIRE(mg_aligned16(wt) && mg_aligned16(bias) && mg_aligned16(y), "mgnns_stem_conv7_fwd: operands must be 16-byte aligned");
    if (B == 0) return 0;
    const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
    dim3 grid((OW + ST_COLS - 1) / ST_COLS, (OH + ST_ROWS - 1) / ST_ROWS, B);
    hipLaunchKernelGGL(stem_conv7_kernel, grid, dim3(256), 0, (hipStream_t)stream, img, H, W, OH, OW,
                       reinterpret_cast<const unsigned short*>(wt), bias, reinterpret_cast<unsigned short*>(y));
    MG_CHECK_LAUNCH("mgnns_stem_conv7_fwd");
    return 0;
}

extern "C" int mgnns_maxpool3x3s2_nhwc_fwd(const void* x, int B, int H, int W, int C, void* y, mgnns_stream_t stream) {
    MG_REQUIRE(x && y, "mgnns_maxpool3x3s2_nhwc_fwd: null pointer");
    MG_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "mgnns_maxpool3x3s2_nhwc_fwd: need C %% 8 == 0 (B=%d H=%d W=%d C=%d)", B, H, W, C);
    MG_REQUIRE(mg_aligned16(x) && mg_aligned16(y), "mgnns_maxpool3x3s2_nhwc_fwd: operands must be 16-byte aligned");
    if (B == 0) return 0;
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)B * OH * OW * (C / 8);
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned short*>(x), B, H, W, C, OH, OW, reinterpret_cast<unsigned short*>(y));
    MG_CHECK_LAUNCH("mgnns_maxpool3x3s2_nhwc_fwd");
    return 0;
}

extern "C" int mgnns_conv_bf16_nhwc_fwd(const void* x, int B, int H, int W, int Cin, const void* wt, const float* bias, int Cout,
                                        int KH, int KW, int stride, int pad, const void* residual, int relu, int out_nchw_f32,
                                        void* y, mgnns_stream_t stream) {
    MG_REQUIRE(x && wt && bias && y, "mgnns_conv_bf16_nhwc_fwd: null pointer");
    MG_REQUIRE(B >= 0 && H > 0 && W > 0, "mgnns_conv_bf16_nhwc_fwd: bad dims B=%d H=%d W=%d", B, H, W);
    MG_REQUIRE(Cin >= 64 && (Cin & (Cin - 1)) == 0, "mgnns_conv_bf16_nhwc_fwd: C_in must be a power of two >= 64 (got %d)", Cin);
    MG_REQUIRE(Cout > 0 && Cout % 8 == 0, "mgnns_conv_bf16_nhwc_fwd: C_out %% 8 != 0 (got %d)", Cout);
    MG_REQUIRE((KH == 1 && KW == 1) || (KH == 3 && KW == 3), "mgnns_conv_bf16_nhwc_fwd: kernel must be 1x1 or 3x3 (got %dx%d)", KH, KW);
    MG_REQUIRE(stride >= 1 && pad >= 0 && pad <= KH / 2, "mgnns_conv_bf16_nhwc_fwd: bad stride %d / padding %d", stride, pad);
    MG_REQUIRE(mg_aligned16(x) && mg_aligned16(wt) && mg_aligned16(bias) && mg_aligned16(y) && (!residual || mg_aligned16(residual)),
               "mgnns_conv_bf16_nhwc_fwd: operands must be 16-byte aligned");
    if (B == 0) return 0;
    ConvArgs a;
    a.x = reinterpret_cast<const unsigned short*>(x);
    a.wt = reinterpret_cast<const unsigned short*>(wt);
    a.bias = bias;
    a.res = reinterpret_cast<const unsigned short*>(residual);
    a.y = y;
    a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
    a.cin_shift = __builtin_ctz((unsigned)Cin);
    a.OH = (H + 2 * pad - KH) / stride + 1;
    a.OW = (W + 2 * pad - KW) / stride + 1;
    MG_REQUIRE(a.OH > 0 && a.OW > 0, "mgnns_conv_bf16_nhwc_fwd: empty output");
    const long long M = (long long)B * a.OH * a.OW;
    MG_REQUIRE(M < (1ll << 31) - TM, "mgnns_conv_bf16_nhwc_fwd: B*OH*OW = %lld does not fit 31 bits", M);
    const long long ybytes = M * Cout * (out_nchw_f32 ? 4 : 2);
    MG_REQUIRE(ybytes < 0xFFFFFFF0ll, "mgnns_conv_bf16_nhwc_fwd: output of %lld bytes exceeds the 4 GiB buffer-store range; split the batch", ybytes);
    MG_REQUIRE((long long)a.OH * a.OW + TM < (1 << 24), "mgnns_conv_bf16_nhwc_fwd: OH*OW = %d too large", a.OH * a.OW);
    a.M = (int)M;
    a.ybytes = (unsigned int)ybytes;
    a.inv_ohw = 1.0f / (float)(a.OH * a.OW);
    a.inv_ow = 1.0f / (float)a.OW;
    a.relu = relu ? 1 : 0;
    const int n_cu = mg_cu_count();
    if (n_cu <= 0) return MGNNS_ERR_LAUNCH;
    const int rc = Cout <= 64 ? launch_conv<2>(a, out_nchw_f32 != 0, (hipStream_t)stream, n_cu)
                              : launch_conv<4>(a, out_nchw_f32 != 0, (hipStream_t)stream, n_cu);
    if (rc) return rc;
    MG_CHECK_LAUNCH("mgnns_conv_bf16_nhwc_fwd");
    return 0;
}
