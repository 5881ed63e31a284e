// Issue rate of bf16 MFMA shapes on one SIMD (s_memtime around a stream of independent MFMAs): hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 2: v_mfma_f32_32x32x16_bf16, 4 independent accumulators (64 registers)
__global__ __launch_bounds__(512) void k32(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    bf16x8 a8, b8;
    for (int j = 0; j < 8; ++j) { a8[j] = (__bf16)(float)(threadIdx.x + j); b8[j] = (__bf16)(float)(j + 1); }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[i], 0, 0, 0);
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a8, b8;
    s16x4 a4, b4;
    for (int j = 0; j < 8; ++j) { a8[j] = (__bf16)(float)(threadIdx.x + j); b8[j] = (__bf16)(float)(j + 1); }
    for (int j = 0; j < 4; ++j) { a4[j] = (short)(threadIdx.x + j); b4[j] = (short)(0x3f80 + j); }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 512 * 4 * 4); hipMalloc(&cyc, 64 * 8);
    unsigned long long h[8];
    const int iters = 2000;
    for (int threads : {256, 512}) {
        for (int mode = 0; mode < 2; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
                else hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
            printf("%s waves/SIMD=%d: %.2f ticks per MFMA per wave (wave 0), %.2f per SIMD-MFMA\n", mode ? "16x16x16" : "16x16x32", threads / 256,
                   (double)h[0] / (iters * 8), (double)h[0] / (iters * 8) / (threads / 256));
        }
    }
    for (int threads : {256, 512}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k32, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("32x32x16 waves/SIMD=%d: %.2f ticks per MFMA per wave (wave 0), %.2f per SIMD-MFMA = %.2f per 16x16x32-equivalent\n", threads / 256,
               (double)h[0] / (iters * 4), (double)h[0] / (iters * 4) / (threads / 256), (double)h[0] / (iters * 4) / (threads / 256) / 2);
    }
    return 0;
}
