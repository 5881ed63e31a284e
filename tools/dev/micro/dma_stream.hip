// Feasibility of an LDS-DMA-fed image-bank kernel: one 512-thread workgroup per sample streams its 2048 x 784-B map through a
// ring of RING slices of BKR rows (global -> LDS by buffer_load ... lds, no VGPRs), every wave reads its share of each landed
// slice back from LDS (ds_read_b128) and takes a max.  Reports TB/s for the whole map.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BKR, int RING, int READBACK, int MODE = 0>
__global__ __launch_bounds__(512) void k(const float* __restrict__ feat, float* __restrict__ out, int K, int P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SLICE = BKR * 784;                       // bytes (P = 196)
    constexpr int PIECES = (SLICE + 1023) / 1024;          // 1-KiB DMA pieces per slice
    constexpr int PPW = (PIECES + 7) / 8;                  // per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t sample_bytes = (size_t)K * P * 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(feat) + (size_t)blockIdx.x * K * P, 0, (int)sample_bytes, 0x00027000);
    const int nslice = K / BKR;
    auto issue = [&](int c) {                              // this wave's pieces of slice c (dummy beyond the end: out of range -> zeros)
        const int slot = c % RING;
        if (MODE == 1) {                                   // the kernel's round-3 form: ROWS of 784 B (49 lanes), waves 4-7 only, BKR / 4 each
            if (wave >= 4 && lane < 49) {
#pragma unroll
                for (int j = 0; j < BKR / 4; ++j) {
                    const int r = (wave - 4) * (BKR / 4) + j;
                    const int off = c < nslice ? (c * BKR + r) * 784 + lane * 16 : 0x7ffffff0;
                    unsigned char* dst = smem + (size_t)slot * (PIECES * 1024) + r * 784;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16, off, 0, 0, 2);
                }
            }
            return;
        }
        if (MODE == 2) {                                   // 1-KiB pieces, waves 4-7 only
            if (wave >= 4) {
#pragma unroll
                for (int j = 0; j < (PIECES + 3) / 4; ++j) {
                    const int pc = (wave - 4) + 4 * j;
                    const int off = c < nslice && pc < PIECES ? c * SLICE + pc * 1024 + lane * 16 : 0x7ffffff0;
                    unsigned char* dst = smem + (size_t)slot * (PIECES * 1024) + (pc < PIECES ? pc : 0) * 1024;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16, off, 0, 0, 2);
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int pc = wave + 8 * j;
            const int off = c < nslice && pc < PIECES ? c * SLICE + pc * 1024 + lane * 16 : 0x7ffffff0;
            unsigned char* dst = smem + (size_t)slot * (PIECES * 1024) + (pc < PIECES ? pc : 0) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16, off, 0, 0, 2);
        }
    };
#pragma unroll
    for (int c = 0; c < RING - 1; ++c) issue(c);
    f32x4 m = {-1e30f, -1e30f, -1e30f, -1e30f};
    for (int c = 0; c < nslice; ++c) {
        issue(c + RING - 1);
        // slice c has landed when at most (RING - 1) * PPW younger pieces are outstanding
        if (MODE == 1) { if (wave >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 1) * (BKR / 4)) : "memory"); }
        else if (MODE == 2) { if (wave >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 1) * ((PIECES + 3) / 4)) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 1) * PPW) : "memory");
        __builtin_amdgcn_s_barrier();
        if (READBACK) {
            const unsigned char* src = smem + (size_t)(c % RING) * (PIECES * 1024);
            for (int i = tid; i < SLICE / 16; i += 512) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src + i * 16);
                m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
            }
        }
        __builtin_amdgcn_s_barrier();                      // the slot is free for slice c + RING (issued next iteration)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[(size_t)blockIdx.x * 512 + tid] = m[0] + m[1] + m[2] + m[3];
}

template <int BKR, int RING, int RB, int MODE = 0>
void run(const float* feat, float* out, int B, size_t stride_f, int nbuf) {
    constexpr int PIECES = (BKR * 784 + 1023) / 1024;
    const size_t lds = (size_t)RING * PIECES * 1024;
    (void)hipFuncSetAttribute((const void*)k<BKR, RING, RB, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<BKR, RING, RB, MODE>), dim3(B), dim3(512), lds, 0, feat + (i % nbuf) * stride_f, out, 2048, 196);
    (void)hipEventRecord(a);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<BKR, RING, RB, MODE>), dim3(B), dim3(512), lds, 0, feat + (i % nbuf) * stride_f, out, 2048, 196);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)B * 2048 * 784;
    printf("mode=%d BK=%3d ring=%d readback=%d LDS=%3zu KB: %.1f us  %.2f TB/s (%s)\n", MODE, BKR, RING, RB, lds / 1024, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12,
           hipGetErrorString(hipGetLastError()));
}

int main() {
    const int B = 256, nbuf = 2;
    const size_t stride_f = (size_t)B * 2048 * 196;
    float *feat, *out;
    (void)hipMalloc(&feat, stride_f * 4 * nbuf);
    (void)hipMalloc(&out, (size_t)B * 512 * 4);
    (void)hipMemset(feat, 0, stride_f * 4 * nbuf);
    run<32, 4, 0>(feat, out, B, stride_f, nbuf);
    run<32, 4, 1>(feat, out, B, stride_f, nbuf);
    run<32, 5, 1>(feat, out, B, stride_f, nbuf);
    run<32, 6, 1>(feat, out, B, stride_f, nbuf);
    run<64, 3, 1>(feat, out, B, stride_f, nbuf);
    run<16, 8, 1>(feat, out, B, stride_f, nbuf);
    run<16, 10, 1>(feat, out, B, stride_f, nbuf);
    run<32, 4, 1, 1>(feat, out, B, stride_f, nbuf);
    run<32, 5, 1, 1>(feat, out, B, stride_f, nbuf);
    run<32, 4, 1, 2>(feat, out, B, stride_f, nbuf);
    run<32, 5, 1, 2>(feat, out, B, stride_f, nbuf);
    run<32, 4, 0, 1>(feat, out, B, stride_f, nbuf);
    run<32, 4, 0, 2>(feat, out, B, stride_f, nbuf);
    return 0;
}
