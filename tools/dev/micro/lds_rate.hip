// ds_read_b128 rate of one CU and of the chip, beside MI355X_MICROARCH.md's 256 B/clk/CU (4 LDS cycles per
// wave-instruction):
//   PAT 0  contiguous 16 B per lane (lane * 16): the guide's conflict-free case
//   PAT 1  the attention core's bank-fragment read: row (lane & 15) of a 672-B-stride tile, 16-B chunk (lane >> 4) + 4 * kstep
// alone (R reads per s_waitcnt), and co-issued with MFMAs at the core's ratio: 13 reads feed 26 v_mfma_f32_16x16x32_bf16 per k-step
// (MF = 1: the reads of k-step i + 1 issued BEFORE the MFMAs of k-step i -- all 13 in flight; MF = 2: one read behind every second MFMA).
// Cycles by s_memtime (wave 0 of workgroup 0) and wall clock by hipEvents over the whole grid.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int PAT, int R, int MF>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < 208 * 42; i += blockDim.x) lds[i] = uint4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint4* base = PAT == 0 ? lds + lane : lds + (lane & 15) * 42 + (lane >> 4);
    constexpr int TSTR = PAT == 0 ? 64 : 16 * 42;            // uint4 between the R reads of a group (PAT 1: the next 16-row tile)
    uint4 v[R];
    f32x4 acc[R][2];
    for (int i = 0; i < R; ++i) { acc[i][0] = acc[i][1] = f32x4{0, 0, 0, 0}; v[i] = base[i * TSTR]; }
    bf16x8 b0, b1;
    for (int i = 0; i < 8; ++i) { b0[i] = (__bf16)1.0f; b1[i] = (__bf16)2.0f; }
    unsigned fold = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int ks = PAT == 0 ? (it & 7) * 64 : (it % 10) * 4;
        if (MF == 0) {                                       // reads only
#pragma unroll
            for (int i = 0; i < R; ++i) fold ^= v[i].x ^ v[i].w;
#pragma unroll
            for (int i = 0; i < R; ++i) v[i] = base[i * TSTR + ks];
        } else if (MF == 1) {                                // all R reads of the next k-step first, then the 2 R MFMAs of this one
            uint4 nv[R];
#pragma unroll
            for (int i = 0; i < R; ++i) nv[i] = base[i * TSTR + ks];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const bf16x8 av = __builtin_bit_cast(bf16x8, v[i]);
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, av, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, av, acc[i][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < R; ++i) v[i] = nv[i];
        } else {                                             // the kernel's form: the refill of tile i right behind its two MFMAs
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const bf16x8 av = __builtin_bit_cast(bf16x8, v[i]);
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, av, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, av, acc[i][1], 0, 0, 0);
                v[i] = base[i * TSTR + ks];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    float s = (float)fold;
    for (int i = 0; i < R; ++i) s += acc[i][0][0] + acc[i][1][3] + (float)v[i].y;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xffff] = s;
    if (lane == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int PAT, int R, int MF>
void run(float* out, unsigned long long* cyc, int waves, int grid) {
    const int iters = 2000;
    const size_t lds = 208 * 42 * 16;
    (void)hipFuncSetAttribute((const void*)k<PAT, R, MF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<PAT, R, MF>), dim3(grid), dim3(waves * 64), lds, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<PAT, R, MF>), dim3(grid), dim3(waves * 64), lds, 0, out, cyc, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long h[16];
    (void)hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
    const double c = (double)h[0] / iters;                   // ticks per group of R reads (wave 0)
    const double bytes_cu = (double)waves * R * 1024;        // per group, whole CU
    printf("%s R=%2d %-22s waves/CU %2d grid %3d: %7.1f ticks per %2d reads (+%2d MFMAs; floor %3d)  %6.1f B/tick/CU  | wall %7.1f us -> %6.1f TB/s chip, %5.1f B/clk/CU at 2.4 GHz\n",
           PAT ? "core 672-B rows" : "contiguous     ", R, MF == 0 ? "reads only" : MF == 1 ? "13 reads, then MFMAs" : "read behind its MFMAs", waves, grid, c, R,
           MF ? 2 * R : 0, MF ? 2 * R * 16 * ((waves + 3) / 4) : R * 4 * waves, bytes_cu / c,
           ms * 1e3, bytes_cu * iters * grid / (ms * 1e-3) / 1e12, bytes_cu * iters / (ms * 1e-3) / 2.4e9);
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 65536 * 4); (void)hipMalloc(&cyc, 128);
    for (int grid : {1, 256}) {
        printf("== ds_read_b128 alone, grid %d ==\n", grid);
        for (int waves : {1, 4, 8, 16}) run<0, 16, 0>(out, cyc, waves, grid);
        for (int waves : {4, 8, 16}) run<1, 13, 0>(out, cyc, waves, grid);
        for (int waves : {4, 8}) run<1, 8, 0>(out, cyc, waves, grid);
        printf("== the core's k-step (13 reads : 26 MFMAs), grid %d ==\n", grid);
        for (int waves : {4, 8}) {
            run<1, 13, 2>(out, cyc, waves, grid);
            run<1, 13, 1>(out, cyc, waves, grid);
        }
    }
    return 0;
}
