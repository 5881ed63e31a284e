// MFMA issue rate with ds_read_b128 refills interleaved (one read per two MFMAs, the attention core's k-step), per waves/CU.
// hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>     // 0: MFMA only; 1: + ds_read_b128 per 2 MFMAs (fragment used a round later); 2: reads only
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < 208 * 42; i += blockDim.x) lds[i] = uint4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint4* a_base = lds + (lane & 15) * 42 + (lane >> 4);
    f32x4 acc[13][2];
    uint4 ga[13];
    for (int i = 0; i < 13; ++i) { acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; ga[i] = a_base[i * 16 * 42]; }
    bf16x8 b0, b1;
    for (int j = 0; j < 8; ++j) { b0[j] = (__bf16)(float)(lane + j); b1[j] = (__bf16)(float)(j + 1); }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int ks = (it % 10) * 4;
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            const bf16x8 av = __builtin_bit_cast(bf16x8, ga[i]);
            if (MODE != 2) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, av, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, av, acc[i][1], 0, 0, 0);
            } else {
                acc[i][0][0] += __builtin_bit_cast(float, ga[i].x);
            }
            if (MODE != 0) ga[i] = a_base[i * 16 * 42 + ks];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 13; ++i) s += acc[i][0][0] + acc[i][1][3];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 512 * 4 * 4); (void)hipMalloc(&cyc, 64 * 8);
    unsigned long long h[8];
    const int iters = 1000;
    const size_t lds = 208 * 42 * 16;
    (void)hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int threads : {64, 256, 512}) {
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), lds, 0, out, cyc, iters);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), lds, 0, out, cyc, iters);
                else hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), lds, 0, out, cyc, iters);
                (void)hipDeviceSynchronize();
            }
            (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
            printf("%s waves/CU=%d: %.1f ticks per k-step (26 MFMAs + 13 reads) wave 0, last wave %.1f\n",
                   mode == 0 ? "mfma only " : mode == 1 ? "mfma + lds " : "lds only  ", threads / 64, (double)h[0] / iters,
                   (double)h[threads / 64 - 1] / iters);
        }
    }
    return 0;
}
