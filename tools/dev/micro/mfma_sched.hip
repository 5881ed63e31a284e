// How should one wave interleave ds_read_b128 refills with v_mfma_f32_32x32x16_bf16 so that the reads hide in the MFMA shadow?
// One k-step = 7 MFMAs (7 row tiles) + 7 refills of an 8-deep fragment ring; patterns:
//   0: M only (no reads)                         1: [M R] x 7, one counted wait per MFMA
//   2: [M M R R] (the kernel's form, wait per 2)  3: [M R] x 7 without any wait (timing only)
//   4: [R M] -- the refill issued BEFORE the MFMA that frees... (reads one slot further ahead)   5: [M M M M R R R R] wait per 4
//   6: pattern 2 with s_setprio 1 around the MFMAs  7: pattern 1 with the wait AFTER the MFMA (for the next one)
// hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int OFF> __device__ __forceinline__ u32x4 rd(unsigned a) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
    return v;
}
template <int N> __device__ __forceinline__ void wt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
#define SB __builtin_amdgcn_sched_barrier(0)
#define MF(i, s) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, __builtin_bit_cast(bf16x8, ga[s]), acc[i], 0, 0, 0)

// patterns 8 / 9: pattern 2 plus the weight stream -- one 1-KiB fragment per k-step through a buffer resource into a 3-deep ring
// (8: requested at the end of the k-step, 9: in its middle, 10: at the end, consumed through vmcnt(0) i.e. ring of 1)
template <int P>
__global__ __launch_bounds__(512) void kw(const uint4* __restrict__ W, float* out, unsigned long long* cyc, int iters) {
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < 224 * 39; i += blockDim.x) lds[i] = uint4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned a0 = (unsigned)(uintptr_t)(lds + (lane & 31) * 39 + (lane >> 5));
    const unsigned a1 = a0 + 4 * 32 * 39 * 16;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(W), 0, 0x7fffffff, 0x00027000);
    f32x16 acc[7];
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    uint4 bq[3];
    for (int d = 0; d < 3; ++d) bq[d] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (wave * 24 + d) * 1024, 0));
    u32x4 ga[8];
    ga[0] = rd<0>(a0); ga[1] = rd<19968>(a0); ga[2] = rd<39936>(a0); ga[3] = rd<59904>(a0);
    ga[4] = rd<0>(a1); ga[5] = rd<19968>(a1); ga[6] = rd<39936>(a1); ga[7] = rd<32>(a0);
    wt<0>();
    SB;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24; ++u) {
#define S(i) ((u * 7 + (i)) & 7)
            const bf16x8 w = __builtin_bit_cast(bf16x8, bq[u % 3]);
            const int fo = ((it & 7) * 8 + wave) * 24 * 1024 + u * 1024;
            wt<6>(); SB; MF(0, S(0)); MF(1, S(1)); SB; ga[S(0)] = rd<64>(a0); ga[S(1)] = rd<19968 + 64>(a0); SB;
            wt<6>(); SB; MF(2, S(2)); MF(3, S(3)); SB; ga[S(2)] = rd<39936 + 64>(a0); ga[S(3)] = rd<59904 + 64>(a0); SB;
            if (P == 9) { bq[u % 3] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, fo, 0)); SB; }
            wt<6>(); SB; MF(4, S(4)); MF(5, S(5)); SB; ga[S(4)] = rd<64>(a1); ga[S(5)] = rd<19968 + 64>(a1); SB;
            wt<7>(); SB; MF(6, S(6)); SB; ga[S(6)] = rd<39936 + 64>(a1); SB;
            if (P == 8) { bq[u % 3] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, fo, 0)); SB; }
#undef S
        }
    }
    wt<0>();
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 7; ++i) s += acc[i][0] + acc[i][15];
    for (int i = 0; i < 8; ++i) s += (float)ga[i][0];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(blockIdx.x * blockDim.x + threadIdx.x) % 2048] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int P>
void runw(const uint4* W, float* out, unsigned long long* cyc, int threads, int grid) {
    const int iters = 64;
    const size_t lds = 224 * 39 * 16;
    (void)hipFuncSetAttribute((const void*)kw<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    unsigned long long h[8];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((kw<P>), dim3(grid), dim3(threads), lds, 0, W, out, cyc, iters);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("pattern %d (W stream) waves/CU=%d grid=%3d: %.1f ticks per k-step (7 MFMAs, floor 224) wave 0, last wave %.1f\n", P, threads / 64, grid,
           (double)h[0] / (iters * 24), (double)h[threads / 64 - 1] / (iters * 24));
}

template <int P>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < 224 * 39; i += blockDim.x) lds[i] = uint4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned a0 = (unsigned)(uintptr_t)(lds + (lane & 31) * 39 + (lane >> 5));
    const unsigned a1 = a0 + 4 * 32 * 39 * 16;
    f32x16 acc[7];
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    bf16x8 w;
    for (int j = 0; j < 8; ++j) w[j] = (__bf16)(float)(lane + j);
    u32x4 ga[8];
    ga[0] = rd<0>(a0); ga[1] = rd<19968>(a0); ga[2] = rd<39936>(a0); ga[3] = rd<59904>(a0);
    ga[4] = rd<0>(a1); ga[5] = rd<19968>(a1); ga[6] = rd<39936>(a1); ga[7] = rd<32>(a0);
    wt<0>();
    SB;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // ring slot of tile i in this k-step: (it*7 + i) % 8 -- unrolled by 8 k-steps so that slots are compile-time
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#define S(i) ((u * 7 + (i)) & 7)
            if (P == 0) {
                MF(0, S(0)); MF(1, S(1)); MF(2, S(2)); MF(3, S(3)); MF(4, S(4)); MF(5, S(5)); MF(6, S(6)); SB;
            } else if (P == 1 || P == 3) {
                if (P == 1) wt<7>(); SB; MF(0, S(0)); SB; ga[S(0)] = rd<64>(a0); SB;
                if (P == 1) wt<7>(); SB; MF(1, S(1)); SB; ga[S(1)] = rd<19968 + 64>(a0); SB;
                if (P == 1) wt<7>(); SB; MF(2, S(2)); SB; ga[S(2)] = rd<39936 + 64>(a0); SB;
                if (P == 1) wt<7>(); SB; MF(3, S(3)); SB; ga[S(3)] = rd<59904 + 64>(a0); SB;
                if (P == 1) wt<7>(); SB; MF(4, S(4)); SB; ga[S(4)] = rd<64>(a1); SB;
                if (P == 1) wt<7>(); SB; MF(5, S(5)); SB; ga[S(5)] = rd<19968 + 64>(a1); SB;
                if (P == 1) wt<7>(); SB; MF(6, S(6)); SB; ga[S(6)] = rd<39936 + 64>(a1); SB;
            } else if (P == 2 || P == 6) {
                wt<6>(); SB; if (P == 6) __builtin_amdgcn_s_setprio(1); MF(0, S(0)); MF(1, S(1)); if (P == 6) __builtin_amdgcn_s_setprio(0); SB; ga[S(0)] = rd<64>(a0); ga[S(1)] = rd<19968 + 64>(a0); SB;
                wt<6>(); SB; if (P == 6) __builtin_amdgcn_s_setprio(1); MF(2, S(2)); MF(3, S(3)); if (P == 6) __builtin_amdgcn_s_setprio(0); SB; ga[S(2)] = rd<39936 + 64>(a0); ga[S(3)] = rd<59904 + 64>(a0); SB;
                wt<6>(); SB; if (P == 6) __builtin_amdgcn_s_setprio(1); MF(4, S(4)); MF(5, S(5)); if (P == 6) __builtin_amdgcn_s_setprio(0); SB; ga[S(4)] = rd<64>(a1); ga[S(5)] = rd<19968 + 64>(a1); SB;
                wt<7>(); SB; MF(6, S(6)); SB; ga[S(6)] = rd<39936 + 64>(a1); SB;
            } else if (P == 5) {
                wt<4>(); SB; MF(0, S(0)); MF(1, S(1)); MF(2, S(2)); MF(3, S(3)); SB;
                ga[S(0)] = rd<64>(a0); ga[S(1)] = rd<19968 + 64>(a0); ga[S(2)] = rd<39936 + 64>(a0); ga[S(3)] = rd<59904 + 64>(a0); SB;
                wt<5>(); SB; MF(4, S(4)); MF(5, S(5)); MF(6, S(6)); SB;
                ga[S(4)] = rd<64>(a1); ga[S(5)] = rd<19968 + 64>(a1); ga[S(6)] = rd<39936 + 64>(a1); SB;
            } else if (P == 7) {
                MF(0, S(0)); SB; ga[S(0)] = rd<64>(a0); wt<7>(); SB;
                MF(1, S(1)); SB; ga[S(1)] = rd<19968 + 64>(a0); wt<7>(); SB;
                MF(2, S(2)); SB; ga[S(2)] = rd<39936 + 64>(a0); wt<7>(); SB;
                MF(3, S(3)); SB; ga[S(3)] = rd<59904 + 64>(a0); wt<7>(); SB;
                MF(4, S(4)); SB; ga[S(4)] = rd<64>(a1); wt<7>(); SB;
                MF(5, S(5)); SB; ga[S(5)] = rd<19968 + 64>(a1); wt<7>(); SB;
                MF(6, S(6)); SB; ga[S(6)] = rd<39936 + 64>(a1); wt<7>(); SB;
            }
#undef S
        }
    }
    wt<0>();
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 7; ++i) s += acc[i][0] + acc[i][15];
    for (int i = 0; i < 8; ++i) s += (float)ga[i][0];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(blockIdx.x * blockDim.x + threadIdx.x) % 2048] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int P>
void run(float* out, unsigned long long* cyc, int threads, int grid) {
    const int iters = 200;
    const size_t lds = 224 * 39 * 16;
    (void)hipFuncSetAttribute((const void*)k<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    unsigned long long h[8];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<P>), dim3(grid), dim3(threads), lds, 0, out, cyc, iters);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("pattern %d waves/CU=%d grid=%3d: %.1f ticks per k-step (7 MFMAs, floor 224) wave 0, last wave %.1f\n", P, threads / 64, grid,
           (double)h[0] / (iters * 8), (double)h[threads / 64 - 1] / (iters * 8));
}

int main() {
    float* out; unsigned long long* cyc; uint4* W;
    (void)hipMalloc(&out, 2048 * 4); (void)hipMalloc(&cyc, 64); (void)hipMalloc(&W, 8 * 8 * 24 * 1024 + 65536);
    (void)hipMemset(W, 0, 8 * 8 * 24 * 1024 + 65536);
    for (int threads : {256, 512}) { runw<8>(W, out, cyc, threads, 256); runw<9>(W, out, cyc, threads, 256); }
    for (int threads : {256, 512}) {
        run<0>(out, cyc, threads, 256); run<1>(out, cyc, threads, 256); run<2>(out, cyc, threads, 256); run<3>(out, cyc, threads, 256);
        run<5>(out, cyc, threads, 256); run<6>(out, cyc, threads, 256); run<7>(out, cyc, threads, 256);
    }
    return 0;
}
