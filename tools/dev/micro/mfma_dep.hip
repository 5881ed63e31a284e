// Issue rate of v_mfma_f32_16x16x32_bf16 when consecutive MFMAs accumulate into the SAME register: N accumulators used round robin
// (N = 1: every MFMA waits for the one before; the split-bf16 kernels issue three products per accumulator with N = 2).
// One wave per SIMD and two; cycles per MFMA by s_memtime.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int N>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)1.0f; }
    f32x4 acc[N];
    for (int i = 0; i < N; ++i) acc[i] = f32x4{0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 12 / N; ++r)
#pragma unroll
            for (int i = 0; i < N; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < N; ++i) s += acc[i][0];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = s;
    if (lane == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int N>
void run(float* out, unsigned long long* cyc, int waves) {
    const int iters = 2000;
    unsigned long long h[8];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<N>), dim3(1), dim3(waves * 64), 0, 0, out, cyc, iters);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%d accumulator(s) round robin, %d wave(s) per SIMD: %.1f cycles per MFMA of wave 0 (last wave %.1f)\n", N, waves / 4,
           (double)h[0] / (iters * 12.0), (double)h[waves - 1] / (iters * 12.0));
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 512 * 4); (void)hipMalloc(&cyc, 64);
    for (int waves : {4, 8}) {
        run<1>(out, cyc, waves); run<2>(out, cyc, waves); run<3>(out, cyc, waves); run<4>(out, cyc, waves); run<6>(out, cyc, waves); run<12>(out, cyc, waves);
    }
    return 0;
}
