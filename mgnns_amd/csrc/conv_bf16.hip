// Row f4: the CNN trunks either side of the path -- torchvision-style ResNet-101 (objects) and ResNet-50
// (Places365) cut after layer4 (MODEL:274-294, 586-595, 629-630), eval mode, as four kernels:
//
//   conv_fold_bn_kernel     one-off weight preparation: BatchNorm (running statistics) folded into the convolution,
//                           w'[o, (kh, kw, c)] = bf16(w[o, c, kh, kw] * gamma[o] / sqrt(var[o] + eps)), b'[o] = beta - mean * scale
//   stem_conv7_kernel       conv1 7x7 / stride 2 / pad 3 (3 -> 64) + bn1 + ReLU, NCHW fp32 image -> NHWC bf16
//   maxpool3x3s2_kernel     MaxPool2d(3, 2, 1) on NHWC bf16
//   conv_igemm_kernel       every 1x1 / 3x3 convolution of the bottlenecks as an implicit GEMM on
//                           v_mfma_f32_16x16x32_bf16:  Y[m, o] = relu?( sum_k X[pixel(m, tap(k)), c(k)] * w'[o, k] + b'[o] + R[m, o] )
//
// Activations are NHWC bf16 ([B, H, W, C], C contiguous = the GEMM's K axis), accumulation fp32.  The implicit GEMM is
// the dense bf16 GEMM of gemm_bf16.hip (256 x 128 workgroup tile, 8 waves, BK = 64, three-stage LDS-DMA ring, XOR chunk
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
    for (int e = tid; e < 3 * ST_PR * ST_PC; e += 256) {
        const int c = e / (ST_PR * ST_PC), rem = e - c * (ST_PR * ST_PC);
        const int r = rem / ST_PC, cc = rem - r * ST_PC;
        const int ir = ir0 + r, ic = ic0 + cc;
        float v = 0.f;
        if ((unsigned)ir < (unsigned)H && (unsigned)ic < (unsigned)W) v = src[((size_t)c * H + ir) * W + ic];
        patch[e] = (unsigned short)f2bf(v);
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
                o.x = f2bf(fmaxf(acc[jt][0] + bv[0], 0.f)) | (f2bf(fmaxf(acc[jt][1] + bv[1], 0.f)) << 16);
                o.y = f2bf(fmaxf(acc[jt][2] + bv[2], 0.f)) | (f2bf(fmaxf(acc[jt][3] + bv[3], 0.f)) << 16);
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
constexpr int TM = 256, BK = 64, NSTAGE = 3, NTHR = 512, NWAVE = NTHR / 64;
constexpr int A_BYTES = TM * BK * 2, A_PIECES = A_BYTES / 1024;      // 32 KB = 32 DMA pieces of 8 rows x 128 B

struct ConvArgs {
    const unsigned short* x;      // [B, H, W, Cin] bf16
    const unsigned short* wt;     // [Cout, KH*KW*Cin] bf16, k = (kh, kw, c)
    const float* bias;            // [Cout]
    const unsigned short* res;    // [M, Cout] bf16 or null
    void* y;                      // [M, Cout] bf16, or [B, Cout, OH*OW] fp32 when out_nchw
    int H, W, Cin, cin_shift, OH, OW, Cout, KH, KW, stride, pad, M, relu, out_nchw;
    int nrb, nct, rps;
};

template <int NJ>
__global__ __launch_bounds__(NTHR) void conv_igemm_kernel(const ConvArgs a) {
    constexpr int TN = 32 * NJ, B_BYTES = TN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int PIECES = STAGE_BYTES / 1024, PPW = PIECES / NWAVE;          // 48 / 6 (NJ = 4), 40 / 5 (NJ = 2)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- XCD-aware tile map (gemm_bf16.hip)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int per = a.rps * a.nct;
    const int sl = j / per, within = j - sl * per;
    const int rb = (sl * 8 + xcd) * a.rps + within / a.nct, ct = within % a.nct;
    if (rb >= a.nrb) return;
    const int m0 = rb * TM, n0 = ct * TN;
    const int wr = wave >> 1, wc = wave & 1;                       // wave tile: rows wr*64.., cols wc*16*NJ..
    const int K = a.KH * a.KW * a.Cin, nk = K / BK;

    // ---- the four A rows this lane feeds (pieces wave + 8 i): input pixel of tap (0, 0) and its element offset
    const int row_in = lane >> 3, slot = lane & 7;
    int ih0[4], iw0[4];
    long long base[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + (wave + NWAVE * i) * 8 + row_in;
        if (m < a.M) {
            const int ohw = a.OH * a.OW;
            const int b = m / ohw, rem = m - b * ohw;
            const int oh = rem / a.OW, ow = rem - oh * a.OW;
            ih0[i] = oh * a.stride - a.pad;
            iw0[i] = ow * a.stride - a.pad;
            base[i] = (((long long)b * a.H + ih0[i]) * a.W + iw0[i]) * a.Cin;
        } else {
            ih0[i] = -(1 << 20);
            iw0[i] = 0;
            base[i] = 0;
        }
    }
    const unsigned short* zsrc = reinterpret_cast<const unsigned short*>(g_zero16);

    auto issue = [&](int kt, int stage) {
        unsigned char* sb = smem + (size_t)stage * STAGE_BYTES;
        const int k0 = kt * BK;
        const int tap = k0 >> a.cin_shift, c0 = k0 & (a.Cin - 1);
        const int kh = a.KW == 3 ? (tap * 11) >> 5 : 0, kw = tap - kh * a.KW;   // KW in {1, 3}
        const int toff = (kh * a.W + kw) * a.Cin + c0;
        auto dma = [&](const unsigned short* src, int p) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(uintptr_t)(sb + (size_t)p * 1024), 16, 0, 0);
        };
        // slot = chunk ^ ((row >> 1) & 7); a piece starts at a multiple of 8 rows: (row >> 1) & 7 = 4 (p & 1) + (row_in >> 1)
#pragma unroll
        for (int i = 0; i < 4; ++i) {                             // A pieces p = wave + 8 i < 32
            const int p = wave + NWAVE * i;
            const int chunk = slot ^ (4 * (p & 1) + (row_in >> 1));
            const bool ok = (unsigned)(ih0[i] + kh) < (unsigned)a.H && (unsigned)(iw0[i] + kw) < (unsigned)a.W;
            dma(ok ? a.x + (base[i] + toff + chunk * 8) : zsrc, p);
        }
#pragma unroll
        for (int i = 4; i < PPW; ++i) {                           // weight pieces
            const int p = wave + NWAVE * i;
            const int chunk = slot ^ (4 * (p & 1) + (row_in >> 1));
            int row = n0 + (p - A_PIECES) * 8 + row_in;
            row = row < a.Cout ? row : a.Cout - 1;
            dma(a.wt + (size_t)row * K + k0 + chunk * 8, p);
        }
    };

    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    auto afrag = [&](const unsigned char* sb, int i, int s) {
        const int row = wr * 64 + i * 16 + fr;
        return *reinterpret_cast<const uint4*>(sb + ((size_t)row * 8 + ((4 * s + fg) ^ ((fr >> 1) & 7))) * 16);
    };
    auto bfrag = [&](const unsigned char* sb, int jj, int s) {
        const int row = wc * 16 * NJ + jj * 16 + fr;
        return *reinterpret_cast<const uint4*>(sb + A_BYTES + ((size_t)row * 8 + ((4 * s + fg) ^ ((fr >> 1) & 7))) * 16);
    };
    uint4 av[2][4], bv[2][NJ];
    auto reads = [&](const unsigned char* sb, int s, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) av[buf][i] = afrag(sb, i, s);
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) bv[buf][jj] = bfrag(sb, jj, s);
    };
    auto mmas = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
                acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bv[buf][jj]),
                                                                    __builtin_bit_cast(bf16x8, av[buf][i]), acc[i][jj], 0, 0, 0);
    };
    // pipeline: see gemm_bf16.hip (mid-iteration bare barrier, waves 0-3 / 4-7 issue their DMA pieces at different points)
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (nk > 2) issue(2, 2);
    reads(smem, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned char* sb = smem + (size_t)(kt % NSTAGE) * STAGE_BYTES;
        reads(sb, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mmas(0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (wave < 4 && kt + 3 < nk) issue(kt + 3, kt % NSTAGE);
            reads(smem + (size_t)((kt + 1) % NSTAGE) * STAGE_BYTES, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        mmas(1);
        __builtin_amdgcn_sched_barrier(0);
        if (wave >= 4 && kt + 1 < nk && kt + 3 < nk) issue(kt + 3, kt % NSTAGE);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue.  acc[i][jj][r] = Y[m0 + wr*64 + 16 i + fr][n0 + wc*16*NJ + 16 jj + 4 fg + r]: a lane holds 4 consecutive
    // channels of a pixel per tile.  One v_permlane16_swap per register between the tiles of a pair (2 jp, 2 jp + 1)
    // trades the odd 16-lane rows of the first with the even rows of the second, after which lane (fr, fg) holds EIGHT
    // consecutive channels -- (fg & 1) * 16 + (fg >> 1) * 8 .. + 7 of the 32-channel pair -- so residual loads and
    // output stores are 16 bytes per lane in 64-byte runs per pixel (the 8-byte form cost as much as the GEMM itself).
    const int ohw = a.OH * a.OW;
    const int ncol = n0 + wc * 16 * NJ + (fg & 1) * 16 + (fg >> 1) * 8;
    uint4 rv[NJ / 2][4];
    if (a.res) {
#pragma unroll
        for (int jp = 0; jp < NJ / 2; ++jp)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + wr * 64 + i * 16 + fr, n = ncol + jp * 32;
                rv[jp][i] = uint4{0u, 0u, 0u, 0u};
                if (m < a.M && n < a.Cout) rv[jp][i] = *reinterpret_cast<const uint4*>(a.res + (size_t)m * a.Cout + n);
            }
    }
#pragma unroll
    for (int jp = 0; jp < NJ / 2; ++jp) {
        const int n = ncol + jp * 32;
        const bool ncok = n < a.Cout;
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
        if (ncok) {
            b0 = *reinterpret_cast<const f32x4*>(a.bias + n);
            b1 = *reinterpret_cast<const f32x4*>(a.bias + n + 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 lo = acc[i][2 * jp], hi = acc[i][2 * jp + 1];
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\t"
                         "v_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7\n\ts_nop 1"
                         : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
            const int m = m0 + wr * 64 + i * 16 + fr;
            if (m >= a.M || !ncok) continue;
            float o[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o[r] = lo[r] + b0[r];
                o[4 + r] = hi[r] + b1[r];
            }
            if (a.res) {
                const unsigned int u[4] = {rv[jp][i].x, rv[jp][i].y, rv[jp][i].z, rv[jp][i].w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[2 * r] += bf2f(u[r] & 0xFFFFu);
                    o[2 * r + 1] += bf2f(u[r] >> 16);
                }
            }
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 8; ++r) o[r] = fmaxf(o[r], 0.f);
            }
            if (a.out_nchw) {
                const int b = m / ohw, p = m - b * ohw;
                float* dst = reinterpret_cast<float*>(a.y) + ((size_t)b * a.Cout + n) * ohw + p;
#pragma unroll
                for (int r = 0; r < 8; ++r) dst[(size_t)r * ohw] = o[r];
            } else {
                uint4 ov;
                ov.x = f2bf(o[0]) | (f2bf(o[1]) << 16);
                ov.y = f2bf(o[2]) | (f2bf(o[3]) << 16);
                ov.z = f2bf(o[4]) | (f2bf(o[5]) << 16);
                ov.w = f2bf(o[6]) | (f2bf(o[7]) << 16);
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(a.y) + (size_t)m * a.Cout + n) = ov;
            }
        }
    }
}

template <int NJ>
int launch_conv(ConvArgs& a, hipStream_t stream) {
    constexpr int TN = 32 * NJ;
    constexpr size_t SMEM = (size_t)NSTAGE * (A_BYTES + TN * BK * 2);
    MG_DYN_LDS(conv_igemm_kernel<NJ>, SMEM);
    a.nrb = (a.M + TM - 1) / TM;
    a.nct = (a.Cout + TN - 1) / TN;
    int rps = 32 / a.nct;
    if (rps < 1) rps = 1;
    if (rps > a.nrb) rps = a.nrb;
    a.rps = rps;
    const int supers = (a.nrb + rps - 1) / rps;
    const int blocks = 8 * ((supers + 7) / 8) * rps * a.nct;
    hipLaunchKernelGGL(conv_igemm_kernel<NJ>, dim3(blocks), dim3(NTHR), SMEM, stream, a);
    return 0;
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
    a.M = (int)M;
    a.relu = relu ? 1 : 0;
    a.out_nchw = out_nchw_f32 ? 1 : 0;
    const int rc = Cout <= 64 ? launch_conv<2>(a, (hipStream_t)stream) : launch_conv<4>(a, (hipStream_t)stream);
    if (rc) return rc;
    MG_CHECK_LAUNCH("mgnns_conv_bf16_nhwc_fwd");
    return 0;
}
