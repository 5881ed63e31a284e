// Whole-chip bf16 MFMA rate (what the 2.5 PF/s "peak" is under sustained load): 256 workgroups x 8 waves (two per SIMD), every wave a
// stream of independent v_mfma_f32_16x16x32_bf16; launches of ~100 / ~400 / ~1600 us.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a8, b8;
    for (int j = 0; j < 8; ++j) { a8[j] = (__bf16)(float)((threadIdx.x + j) & 7); b8[j] = (__bf16)(float)(j + 1); }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    unsigned long long h[256];
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int iters : {1500, 6000, 24000, 96000}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, cyc, iters);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost);
            const double flops = 256.0 * 8 * iters * 8 * 16384.0;
            printf("iters %6d: %.1f us  %.3f PFLOP/s  (wg 0: %.2f s_memtime ticks per MFMA of its SIMD pair => %.2f GHz if an MFMA is 16 cycles)\n", iters, ms * 1e3,
                   flops / (ms * 1e-3) / 1e15, (double)h[0] / (iters * 8.0 * 2), 16.0 * iters * 8 * 2 / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
