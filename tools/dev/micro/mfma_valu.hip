// Do a partner wave's VALU instructions overlap with a wave's MFMA stream on the same SIMD?  Waves 0-3: MFMAs only; waves 4-7
// (same SIMDs): FMAs only, MFMAs only, or idle.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int PARTNER>      // 0 idle, 1 independent v_fma chains, 2 MFMAs too, 3 v_pk_fma
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, (float)lane};
    bf16x8 a8, b8;
    for (int j = 0; j < 8; ++j) { a8[j] = (__bf16)(float)(lane + j); b8[j] = (__bf16)(float)(j + 1); }
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = lane * 0.001f + j;
    const float m = 1.0001f, c = 0.5f;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4 || PARTNER == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
        }
    } else if (PARTNER == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) x[j] = __builtin_fmaf(x[j], m, c);
        }
    } else if (PARTNER == 3) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2* xp = reinterpret_cast<f32x2*>(x);
        const f32x2 m2 = {m, m}, c2 = {c, c};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xp[j] = __builtin_elementwise_fma(xp[j], m2, c2);
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    for (int j = 0; j < 16; ++j) s += x[j];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = s;
    if (lane == 0) cyc[wave] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 2048 * 4); (void)hipMalloc(&cyc, 64);
    unsigned long long h[8];
    const int iters = 2000;
    for (int p = 0; p < 4; ++p) {
        for (int rep = 0; rep < 2; ++rep) {
            if (p == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(512), 0, 0, out, cyc, iters);
            if (p == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(512), 0, 0, out, cyc, iters);
            if (p == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(512), 0, 0, out, cyc, iters);
            if (p == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(512), 0, 0, out, cyc, iters);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("partner %s: MFMA wave %.2f ticks per MFMA; partner wave %.2f ticks per %s\n",
               p == 0 ? "idle" : p == 1 ? "v_fma" : p == 2 ? "mfma" : "v_pk_fma", (double)h[0] / (iters * 8),
               (double)h[4] / (iters * (p == 3 ? 8 : p == 2 ? 8 : 16)), p == 2 ? "MFMA" : p == 3 ? "v_pk_fma" : "v_fma");
    }
    return 0;
}
