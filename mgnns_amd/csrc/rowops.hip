// Row-wise / gather kernels: embedding gather, custom LayerNorm, label-attention core,
// transpose-with-padding (weight layout prep).
#include "common.hpp"

namespace {

// ---- embedding gather: one wave per row, 16-B lanes when D % 4 == 0 -------------------------
__global__ __launch_bounds__(256) void embedding_kernel(const int64_t* __restrict__ idx, int64_t n,
                                                        const float* __restrict__ table, int V, int D,
                                                        float* __restrict__ out, int vec) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t r = row0; r < n; r += stride) {
        int64_t id = idx[r];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);  // clamped; the wrapper validates ranges
        const float* src = table + id * D;
        float* dst = out + r * D;
        if (vec) {
            for (int c = lane; c < D / 4; c += 64)
                reinterpret_cast<f32x4*>(dst)[c] = reinterpret_cast<const f32x4*>(src)[c];
        } else {
            for (int c = lane; c < D; c += 64) dst[c] = src[c];
        }
    }
}

// ---- custom LayerNorm: one wave per row (submodules.py:153-156) ------------------------------
// mean over D, UNBIASED std (divide by D-1), eps added to std.
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int rows, int D,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* xr = x + (size_t)r * D;
    float v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        v[i] = c < D ? xr[c] : 0.f;
        s += v[i];
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        const float d = c < D ? v[i] - mean : 0.f;
        q += d * d;
    }
    const float var = wave_sum(q) / (float)(D - 1);
    const float inv = 1.0f / (sqrtf(var) + eps);
    float* yr = y + (size_t)r * D;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < D) yr[c] = gamma[c] * (v[i] - mean) * inv + beta[c];
    }
}

// ---- label attention core (MODEL:101-131): one wave per (b, l, h), lanes over dh <= 64 ---------
__global__ __launch_bounds__(256) void label_attn_core_kernel(const float* __restrict__ Q,
                                                              const float* __restrict__ K,
                                                              const float* __restrict__ V, int B, int NLQ,
                                                              int n_heads, int dh, float inv_scale,
                                                              float* __restrict__ x, const unsigned char* __restrict__ mask) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t total = (int64_t)B * NLQ * n_heads;
    if (item >= total) return;
    const int h = (int)(item % n_heads);
    const int l = (int)((item / n_heads) % NLQ);
    const int b = (int)(item / ((int64_t)n_heads * NLQ));
    const int hid = n_heads * dh;
    const bool on = lane < dh;
    float e = on ? Q[(size_t)l * hid + h * dh + lane] * K[(size_t)b * hid + h * dh + lane] * inv_scale
                 : -INFINITY;
    // Attention(mask=...): energy.masked_fill(mask == 0, -1e10) behind the scaling (MODEL:118-119); mask is the byte
    // image of the broadcast mask, [B, NLQ, heads * dh]
    if (mask && on && mask[((size_t)b * NLQ + l) * hid + h * dh + lane] == 0) e = -1e10f;
    const float m = wave_max(e);
    const float p = on ? expf(e - m) : 0.f;
    const float z = wave_sum(p);
    if (on) x[((size_t)b * NLQ + l) * hid + h * dh + lane] = (p / z) * V[(size_t)b * hid + h * dh + lane];
}

// ---- out[c*ld + r] = in[r*cols + c], zero padded to ld ------------------------------------------
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* __restrict__ in, int rows, int cols,
                                                            float* __restrict__ out, int ld) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? in[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;   // out row = c, out col = r
        if (c < cols && r < ld) out[(size_t)c * ld + r] = tile[tx][i];
    }
}

// logits[b, n] = bias[n] + sum_p f_p[b, :] . W[n, p*D : (p+1)*D]: the classifier over the four fusion features WITHOUT
// materialising their concatenation (MODEL:560-566); one wave per sample, lane-strided partial sums + DPP reduction
// NI = ceil(D / 64): the sample's four feature rows are requested up front (4 * NI independent loads per lane), every weight
// row likewise -- as a plain loop this kernel was a chain of ~60 dependent global round trips (21 us on an idle chip)
template <int NI>
__global__ __launch_bounds__(256) void classifier_head_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                              const float* __restrict__ f2, const float* __restrict__ f3, int B,
                                                              int D, const float* __restrict__ W, const float* __restrict__ bias,
                                                              int NL, float* __restrict__ logits) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* f[4] = {f0 + (size_t)b * D, f1 + (size_t)b * D, f2 + (size_t)b * D, f3 + (size_t)b * D};
    float x[4][NI];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < NI; ++i) x[p][i] = lane + 64 * i < D ? f[p][lane + 64 * i] : 0.f;
    for (int n = 0; n < NL; ++n) {
        const float* w = W + (size_t)n * 4 * D;
        float wv[4][NI];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < NI; ++i) wv[p][i] = lane + 64 * i < D ? w[p * D + lane + 64 * i] : 0.f;
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < NI; ++i) s = fmaf(x[p][i], wv[p][i], s);
        s = wave_sum(s);
        if (lane == 0) logits[(size_t)b * NL + n] = s + bias[n];
    }
}

// The classifier WITHOUT a launch of its own behind the four fusion stacks: logits = bias + sum_p W[:, pD:(p+1)D] f_p is linear in
// the four features, so each stack's chain ends with ITS part -- one wave per sample writes parts[p][b][:] -- and whichever
// of the nparts launches finishes last adds the parts in index order (deterministic) and writes the logits.  Saves the
// segment boundary in front of the head: dependent work behind events of three other streams started 15-28 us after the last
// of them (profiles/r03_timeline.txt, tools/dev/boundary_gap.py).  Hand-over without fences, like the fused layer kernel: the
// parts leave through system-scope write-through stores (sc0 | sc1), every thread waits for its own acknowledgements
// (vmcnt(0)), ONE relaxed agent-scope atomic per workgroup counts arrivals, the last arriver reads with loads that bypass the
// non-coherent caches and re-arms the counter.
template <int NI>
__global__ __launch_bounds__(256) void classifier_part_kernel(const float* __restrict__ f, int part, int nparts, int B, int D,
                                                              const float* __restrict__ W, const float* __restrict__ bias, int NL,
                                                              float* __restrict__ parts, int* __restrict__ counter,
                                                              float* __restrict__ logits) {
    __shared__ int s_last;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(parts, 0, nparts * B * NL * 4, 0x00027000);
    if (b < B) {
        const float* fb = f + (size_t)b * D;
        float x[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) x[i] = lane + 64 * i < D ? fb[lane + 64 * i] : 0.f;
        for (int n = 0; n < NL; ++n) {
            const float* w = W + (size_t)n * nparts * D + (size_t)part * D;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i) s = fmaf(x[i], lane + 64 * i < D ? w[lane + 64 * i] : 0.f, s);
            s = wave_sum(s);
            if (lane == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, s), prs, ((part * B + b) * NL + n) * 4, 0, 17);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's write-through stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = nparts * (int)gridDim.x;
        const int old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == total - 1;
        if (s_last) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next forward
    }
    __syncthreads();
    if (!s_last) return;
    for (int i = threadIdx.x; i < B * NL; i += 256) {
        float s = 0.f;
        for (int p = 0; p < nparts; ++p)
            s += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, (p * B * NL + i) * 4, 0, 17));
        logits[i] = s + bias[i % NL];
    }
}

}  // namespace

extern "C" int mgnns_classifier_head_fwd(const float* f0, const float* f1, const float* f2, const float* f3, int B, int D,
                                         const float* W, const float* bias, int NL, float* logits, mgnns_stream_t stream) {
    MG_REQUIRE(B >= 0 && D > 0 && NL > 0, "mgnns_classifier_head_fwd: bad dims B=%d D=%d NL=%d", B, D, NL);
    if (B == 0) return 0;
    MG_REQUIRE(f0 && f1 && f2 && f3 && W && bias && logits, "mgnns_classifier_head_fwd: null pointer");
    MG_REQUIRE(D <= 1024, "mgnns_classifier_head_fwd: feature width %d unsupported (<= 1024)", D);
    const dim3 grid((B + 3) / 4), blk(256);
    if (D <= 320)
        hipLaunchKernelGGL(classifier_head_kernel<5>, grid, blk, 0, (hipStream_t)stream, f0, f1, f2, f3, B, D, W, bias, NL, logits);
    else
        hipLaunchKernelGGL(classifier_head_kernel<16>, grid, blk, 0, (hipStream_t)stream, f0, f1, f2, f3, B, D, W, bias, NL, logits);
    MG_CHECK_LAUNCH("mgnns_classifier_head_fwd");
    return 0;
}

extern "C" int mgnns_classifier_part_fwd(const float* f, int part, int nparts, int B, int D, const float* W, const float* bias,
                                         int NL, float* parts, int* counter, float* logits, mgnns_stream_t stream) {
    MG_REQUIRE(B >= 0 && D > 0 && NL > 0 && nparts > 0 && part >= 0 && part < nparts,
               "mgnns_classifier_part_fwd: bad dims B=%d D=%d NL=%d part=%d/%d", B, D, NL, part, nparts);
    if (B == 0) return 0;
    MG_REQUIRE(f && W && bias && parts && counter && logits, "mgnns_classifier_part_fwd: null pointer");
    MG_REQUIRE(D <= 1024, "mgnns_classifier_part_fwd: feature width %d unsupported (<= 1024)", D);
    MG_REQUIRE((double)nparts * B * NL * 4 < 2147483648.0, "mgnns_classifier_part_fwd: parts beyond 2 GiB");
    const dim3 grid((B + 3) / 4), blk(256);
    if (D <= 320)
        hipLaunchKernelGGL(classifier_part_kernel<5>, grid, blk, 0, (hipStream_t)stream, f, part, nparts, B, D, W, bias, NL, parts, counter, logits);
    else
        hipLaunchKernelGGL(classifier_part_kernel<16>, grid, blk, 0, (hipStream_t)stream, f, part, nparts, B, D, W, bias, NL, parts, counter, logits);
    MG_CHECK_LAUNCH("mgnns_classifier_part_fwd");
    return 0;
}

// head-difference term of a multi-head attention output (diff_outputs, submodules.py:38-52, is_regu=True): per sample the mean
// over the ordered head pairs i != j of cos^2(o_i, o_j), o_h = the head's output vector [dv] (F.normalize: x / max(|x|, 1e-12)).
// One wave per sample: the H head norms and the H (H - 1) / 2 dot products are wave reductions over dv.
template <int HMAX>
__global__ __launch_bounds__(256) void head_diff_kernel(const float* __restrict__ o, int B, int H, int dv, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* ob = o + (size_t)b * H * dv;
    float inv[HMAX];
#pragma unroll
    for (int h = 0; h < HMAX; ++h) {
        float s = 0.f;
        if (h < H)
            for (int d = lane; d < dv; d += 64) s = fmaf(ob[h * dv + d], ob[h * dv + d], s);
        inv[h] = 1.0f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < HMAX; ++i)
#pragma unroll
        for (int j = i + 1; j < HMAX; ++j) {
            if (j < H) {
                float s = 0.f;
                for (int d = lane; d < dv; d += 64) s = fmaf(ob[i * dv + d] * inv[i], ob[j * dv + d] * inv[j], s);
                const float c = wave_sum(s);
                acc += 2.0f * c * c;                      // (i, j) and (j, i)
            }
        }
    // n_head = 1: 0 / 0 = NaN, like the reference's sum(...).div_(0)
    if (lane == 0) out[b] = acc / (float)(H * (H - 1));
}

extern "C" int mgnns_head_diff_fwd(const float* o, int B, int H, int dv, float* out, mgnns_stream_t stream) {
    MG_REQUIRE(o && out, "mgnns_head_diff_fwd: null pointer");
    MG_REQUIRE(B >= 0 && H > 0 && H <= 16 && dv > 0, "mgnns_head_diff_fwd: bad dims B=%d n_head=%d (<= 16) d_v=%d", B, H, dv);
    if (B == 0) return 0;
    hipLaunchKernelGGL(head_diff_kernel<16>, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, o, B, H, dv, out);
    MG_CHECK_LAUNCH("mgnns_head_diff_fwd");
    return 0;
}

extern "C" int mgnns_embedding_fwd(const int64_t* idx, int64_t n, const float* table, int V, int D,
                                   float* out, mgnns_stream_t stream) {
    MG_REQUIRE(idx && table && out, "mgnns_embedding_fwd: null pointer");
    MG_REQUIRE(n >= 0 && V > 0 && D > 0, "mgnns_embedding_fwd: bad dims n=%lld V=%d D=%d", (long long)n, V, D);
    if (n == 0) return 0;
    const int vec = (D % 4 == 0) && mg_aligned16(table) && mg_aligned16(out);
    int64_t blocks = (n + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(embedding_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, idx, n, table,
                       V, D, out, vec);
    MG_CHECK_LAUNCH("mgnns_embedding_fwd");
    return 0;
}

extern "C" int mgnns_layernorm_fwd(const float* x, int rows, int D, const float* gamma, const float* beta,
                                   float eps, float* y, mgnns_stream_t stream) {
    MG_REQUIRE(x && gamma && beta && y, "mgnns_layernorm_fwd: null pointer");
    MG_REQUIRE(rows >= 0 && D > 1 && D <= 1024, "mgnns_layernorm_fwd: D=%d out of range (2..1024)", D);
    if (rows == 0) return 0;
    dim3 grid((rows + 3) / 4);
    if (D <= 320)
        hipLaunchKernelGGL(layernorm_kernel<5>, grid, dim3(256), 0, (hipStream_t)stream, x, rows, D, gamma, beta, eps, y);
    else
        hipLaunchKernelGGL(layernorm_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, x, rows, D, gamma, beta, eps, y);
    MG_CHECK_LAUNCH("mgnns_layernorm_fwd");
    return 0;
}

extern "C" int mgnns_label_attn_core_masked_fwd(const float* Q, const float* K, const float* V, const unsigned char* mask,
                                                int B, int NLQ, int n_heads, int dh, float* x, mgnns_stream_t stream) {
    MG_REQUIRE(Q && K && V && x, "mgnns_label_attn_core_fwd: null pointer");
    MG_REQUIRE(B >= 0 && NLQ > 0 && n_heads > 0 && dh > 0 && dh <= 64,
               "mgnns_label_attn_core_fwd: bad dims B=%d NLQ=%d heads=%d dh=%d (dh<=64)", B, NLQ, n_heads, dh);
    if (B == 0) return 0;
    const int64_t total = (int64_t)B * NLQ * n_heads;
    hipLaunchKernelGGL(label_attn_core_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       Q, K, V, B, NLQ, n_heads, dh, 1.0f / sqrtf((float)dh), x, mask);
    MG_CHECK_LAUNCH("mgnns_label_attn_core_fwd");
    return 0;
}

extern "C" int mgnns_label_attn_core_fwd(const float* Q, const float* K, const float* V, int B, int NLQ,
                                         int n_heads, int dh, float* x, mgnns_stream_t stream) {
    return mgnns_label_attn_core_masked_fwd(Q, K, V, nullptr, B, NLQ, n_heads, dh, x, stream);
}

extern "C" int mgnns_transpose_pad(const float* in, int rows, int cols, float* out, int ld,
                                   mgnns_stream_t stream) {
    MG_REQUIRE(in && out, "mgnns_transpose_pad: null pointer");
    MG_REQUIRE(rows > 0 && cols > 0 && ld >= rows, "mgnns_transpose_pad: bad dims rows=%d cols=%d ld=%d", rows, cols, ld);
    dim3 grid((cols + 31) / 32, (ld + 31) / 32);
    hipLaunchKernelGGL(transpose_pad_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, rows, cols, out, ld);
    MG_CHECK_LAUNCH("mgnns_transpose_pad");
    return 0;
}
