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

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int N> struct IC { static constexpr int v = N; };
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IC<I>{}); static_for<I + 1, N>(f); }
}
// inline asm: the compiler can neither narrow the read to the dwords that are used nor move it; waits are counted by hand
template <int OFF>
__device__ __forceinline__ u32x4 rd128(unsigned a) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

template <int PAT, int R, int MF>
__global__ __launch_bounds__(MF ? 512 : 1024) void k(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < 208 * 42; i += blockDim.x) lds[i] = uint4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // byte address of this lane's 16 bytes of read 0; read i at + i * TSTR (two bases: the immediate offset field is 16 bits)
    constexpr int TSTR = PAT == 0 ? 1024 : 16 * 42 * 16, KSTR = PAT == 0 ? 0 : 64, SPLIT = 7;
    const unsigned a_lo = (unsigned)(uintptr_t)(lds + (PAT == 0 ? lane : (lane & 15) * 42 + (lane >> 4)));
    const unsigned a_hi = a_lo + SPLIT * TSTR;
    auto rd = [&](auto ic, auto kc) {
        constexpr int i = decltype(ic)::v, ks = decltype(kc)::v;
        if constexpr (i < SPLIT) return rd128<i * TSTR + ks * KSTR>(a_lo);
        else return rd128<(i - SPLIT) * TSTR + ks * KSTR>(a_hi);
    };
    u32x4 v[R], nv[MF == 1 ? R : 1];
    f32x4 acc[MF ? R : 1][2];
    for (int i = 0; i < (MF ? R : 1); ++i) acc[i][0] = acc[i][1] = f32x4{0, 0, 0, 0};
    bf16x8 b0, b1;
    for (int i = 0; i < 8; ++i) { b0[i] = (__bf16)1.0f; b1[i] = (__bf16)2.0f; }
    unsigned fold = 0;
    static_for<0, R>([&](auto ic) { v[decltype(ic)::v] = rd(ic, IC<0>{}); });
    lgkm<0>();
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 10) {
        static_for<0, 10>([&](auto kc) {                     // ten k-steps written out (immediate offsets)
            constexpr int ks = decltype(kc)::v, kn = (ks + 1) % 10;
            if constexpr (MF == 0) {                         // reads only: R in flight, one wait
                static_for<0, R>([&](auto ic) { v[decltype(ic)::v] = rd(ic, IC<kn>{}); });
                lgkm<0>();
                static_for<0, R>([&](auto ic) { fold ^= v[decltype(ic)::v].x; });
            } else if constexpr (MF == 1) {                  // ALL R reads of the next k-step first, then this k-step's 2 R MFMAs
                auto& cur = (ks & 1) ? nv : v;
                auto& nxt = (ks & 1) ? v : nv;
                static_for<0, R>([&](auto ic) { nxt[decltype(ic)::v] = rd(ic, IC<kn>{}); });
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, R>([&](auto ic) {
                    constexpr int i = decltype(ic)::v;
                    const bf16x8 av = __builtin_bit_cast(bf16x8, cur[i]);
                    acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, av, acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, av, acc[i][1], 0, 0, 0);
                });
                __builtin_amdgcn_sched_barrier(0);
                lgkm<0>();
                __builtin_amdgcn_sched_barrier(0);
            } else {                                         // the kernel's form: ring of RA, refill right behind the two MFMAs
                constexpr int RA = MF == 2 ? 8 : R;          // MF 2: ring 8 (the kernel's), MF 3: ring 13
                static_for<0, R>([&](auto ic) {
                    constexpr int i = decltype(ic)::v;
                    lgkm<RA - 1>();                          // fragment i landed (RA - 1 younger reads may be in flight)
                    __builtin_amdgcn_sched_barrier(0);
                    const bf16x8 av = __builtin_bit_cast(bf16x8, v[i]);
                    acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, av, acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, av, acc[i][1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // slot of fragment i + RA of the (k-step, tile) sequence
                    constexpr int n2 = i + RA, i2 = n2 % R, k2 = (ks + n2 / R) % 10;
                    if constexpr (RA == R) v[i] = rd(IC<i2>{}, IC<k2>{});
                    else v[i2 < R ? i2 : 0] = rd(IC<i2>{}, IC<k2>{});
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        });
    }
    lgkm<0>();
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    float s = (float)fold;
    for (int i = 0; i < (MF ? R : 1); ++i) s += acc[i][0][0] + acc[i][1][3];
    for (int i = 0; i < R; ++i) s += (float)v[i].y;
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
           PAT ? "core 672-B rows" : "contiguous     ", R, MF == 0 ? "reads only" : MF == 1 ? "13 reads, then MFMAs" : MF == 2 ? "ring 8, refill behind" : "ring 13, refill behind", waves, grid, c, R,
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
            run<1, 13, 3>(out, cyc, waves, grid);
            run<1, 13, 1>(out, cyc, waves, grid);
        }
    }
    return 0;
}
