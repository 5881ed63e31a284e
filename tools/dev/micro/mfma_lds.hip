// MFMA issue rate of the attention core's k-step (26 MFMAs, 13 ds_read_b128 refills, 2 L2-resident weight fragments BD k-steps
// ahead) per waves/CU and per ring depth.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MODE bit 0: LDS refills, bit 1: weight fragment loads; BD: weight ring depth
template <int MODE, int BD>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ W, float* out, unsigned long long* cyc, int iters) {
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < 208 * 42; i += blockDim.x) lds[i] = uint4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint4* a_base = lds + (lane & 15) * 42 + (lane >> 4);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(W), 0, 0x7fffffff, 0x00027000);
    f32x4 acc[13][2];
    uint4 ga[13];
    uint4 bq[BD][2];
    for (int i = 0; i < 13; ++i) { acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; ga[i] = a_base[i * 16 * 42]; }
    for (int d = 0; d < BD; ++d) {
        bq[d][0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (wave * 20 + d) * 1024, 0));
        bq[d][1] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (wave * 20 + 10 + d) * 1024, 0));
    }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += BD) {
#pragma unroll
        for (int u = 0; u < BD; ++u) {
            const int ks = ((it + u) % 10) * 4;
            const int unit = ((it + u) / 10) % 8;
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, bq[u][0]), b1 = __builtin_bit_cast(bf16x8, bq[u][1]);
#pragma unroll
            for (int i = 0; i < 13; ++i) {
                const bf16x8 av = __builtin_bit_cast(bf16x8, ga[i]);
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, av, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, av, acc[i][1], 0, 0, 0);
                if (MODE & 1) ga[i] = a_base[i * 16 * 42 + ks];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE & 2) {
                const int fo = ((unit * 8 + wave) * 20 + (ks >> 2)) * 1024;
                bq[u][0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, fo, 0));
                bq[u][1] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, fo + 10240, 0));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 13; ++i) s += acc[i][0][0] + acc[i][1][3];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(blockIdx.x * blockDim.x + threadIdx.x) % 2048] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}


typedef float f32x16 __attribute__((ext_vector_type(16)));
// The same k-step on v_mfma_f32_32x32x16_bf16: 7 row tiles of 32 (624-B row stride), one ds_read_b128 refill per MFMA, one weight
// fragment per k-step of 16 (KV = 1), or K and V sharing each bank fragment (KV = 2: 14 MFMAs, 7 refills, 2 weight fragments)
template <int MODE, int BD, int KV>
__global__ __launch_bounds__(KV == 2 ? 256 : 512) void k32(const uint4* __restrict__ W, float* out, unsigned long long* cyc, int iters) {
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < 224 * 39; i += blockDim.x) lds[i] = uint4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint4* a_base = lds + (lane & 31) * 39 + (lane >> 5);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(W), 0, 0x7fffffff, 0x00027000);
    f32x16 acc[7][KV];
    uint4 ga[7];
    uint4 bq[BD][KV];
    for (int i = 0; i < 7; ++i) {
        for (int v = 0; v < KV; ++v) for (int j = 0; j < 16; ++j) acc[i][v][j] = 0.f;
        ga[i] = a_base[i * 32 * 39];
    }
    for (int d = 0; d < BD; ++d)
        for (int v = 0; v < KV; ++v)
            bq[d][v] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (wave * 40 + v * 20 + d) * 1024, 0));
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += BD) {
#pragma unroll
        for (int u = 0; u < BD; ++u) {
            const int ks = ((it + u) % 19) * 2;
            const int unit = ((it + u) / 19) % 4;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const bf16x8 av = __builtin_bit_cast(bf16x8, ga[i]);
#pragma unroll
                for (int v = 0; v < KV; ++v)
                    acc[i][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bq[u][v]), av, acc[i][v], 0, 0, 0);
                if (MODE & 1) ga[i] = a_base[i * 32 * 39 + ks];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE & 2) {
#pragma unroll
                for (int v = 0; v < KV; ++v) {
                    const int fo = ((unit * 8 + wave) * 40 + v * 20 + (ks >> 1)) * 1024;
                    bq[u][v] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, fo, 0));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 7; ++i) for (int v = 0; v < KV; ++v) s += acc[i][v][0] + acc[i][v][15];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(blockIdx.x * blockDim.x + threadIdx.x) % 2048] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int MODE, int BD, int KV>
void run32(const uint4* W, float* out, unsigned long long* cyc, int threads, int grid) {
    const int iters = 1140;
    const size_t lds = 224 * 39 * 16;
    (void)hipFuncSetAttribute((const void*)k32<MODE, BD, KV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    unsigned long long h[8];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k32<MODE, BD, KV>), dim3(grid), dim3(threads), lds, 0, W, out, cyc, iters);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("32x32x16 KV=%d lds=%d wfrag=%d BD=%d waves/CU=%d grid=%3d: %.1f ticks per k-step of 16 (%d MFMAs, floor %d) wave 0, last wave %.1f  [per 16x16x32-form k-step of 32, both waves of a SIMD: %.1f]\n",
           KV, MODE & 1, (MODE >> 1) & 1, BD, threads / 64, grid, (double)h[0] / iters, 7 * KV, 7 * KV * 32, (double)h[threads / 64 - 1] / iters,
           (double)h[0] / iters * 2);
}

template <int MODE, int BD>
void run(const uint4* W, float* out, unsigned long long* cyc, int threads, int grid) {
    const int iters = 1000;
    const size_t lds = 208 * 42 * 16;
    (void)hipFuncSetAttribute((const void*)k<MODE, BD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    unsigned long long h[8];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<MODE, BD>), dim3(grid), dim3(threads), lds, 0, W, out, cyc, iters);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("lds=%d wfrag=%d BD=%d waves/CU=%d grid=%3d: %.1f ticks per k-step (26 MFMAs) wave 0, last wave %.1f\n", MODE & 1, (MODE >> 1) & 1, BD,
           threads / 64, grid, (double)h[0] / iters, (double)h[threads / 64 - 1] / iters);
}

int main() {
    float* out; unsigned long long* cyc; uint4* W;
    (void)hipMalloc(&out, 2048 * 4); (void)hipMalloc(&cyc, 64); (void)hipMalloc(&W, 8 * 8 * 20 * 1024 + 65536);
    (void)hipMemset(W, 0, 8 * 8 * 20 * 1024 + 65536);
    for (int grid : {1, 256}) {
        for (int threads : {256, 512}) {
            run<0, 2>(W, out, cyc, threads, grid);
            run<1, 2>(W, out, cyc, threads, grid);
            run<3, 2>(W, out, cyc, threads, grid);
            run<3, 5>(W, out, cyc, threads, grid);
            run<2, 2>(W, out, cyc, threads, grid);
        }
    }
    for (int grid : {1, 256}) {
        for (int threads : {256, 512}) {
            run32<0, 4, 1>(W, out, cyc, threads, grid);
            run32<1, 4, 1>(W, out, cyc, threads, grid);
            run32<3, 4, 1>(W, out, cyc, threads, grid);
            run32<3, 2, 1>(W, out, cyc, threads, grid);
        }
        run32<0, 2, 2>(W, out, cyc, 256, grid);
        run32<1, 2, 2>(W, out, cyc, 256, grid);
        run32<3, 2, 2>(W, out, cyc, 256, grid);
    }
    return 0;
}
