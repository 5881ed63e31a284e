// Does the L2 deliver BYTES or REQUESTS?  The dense GEMM's operand pieces are 16 rows x 64 B (BK = 32 bf16) or 8 rows x 128 B (BK = 64): the
// same 1 KiB per wave-instruction as 16 half lines or 8 whole lines.  Every CU streams an L2-resident [R rows x RS bytes] matrix of its
// XCD k-slice by k-slice (row segments of SEG bytes, row stride RS), LDS-DMA, 8 loader waves x 8 pieces in flight.
// hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int SEG>
__global__ __launch_bounds__(512) void k(const unsigned char* __restrict__ buf, int R, int RS, int RW, int passes, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int LPR = SEG / 16, RPP = 64 / LPR;                  // lanes per row segment, rows per 1-KiB piece
    const unsigned xcd = blockIdx.x & 7, cu = blockIdx.x >> 3;
    const size_t region = (size_t)R * RS;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(buf) + xcd * region, 0, (int)region, 0x00027000);
    const unsigned lane_off = (unsigned)(lane / LPR) * RS + (lane % LPR) * 16;
    const int npiece = R / RPP, nk = RW / SEG;
    unsigned char* slot = smem + (size_t)wave * 8 * 1024;
    int p = (cu * 7 + wave) % npiece, kk = (cu * 3) % nk;
    const int total = passes * (npiece * nk / 8);
    for (int i = 0; i < total; i += 8) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(uintptr_t)(slot + (h * 4 + u) * 1024), 16,
                                                         lane_off + (unsigned)(p * RPP) * RS, kk * SEG, 0, 0);
                p += 8;
                if (p >= npiece) { p -= npiece; kk = kk + 1 == nk ? 0 : kk + 1; }
            }
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[(blockIdx.x * 512 + tid) & 0xffff] = (float)slot[lane];
}

template <int SEG>
void run(const unsigned char* buf, float* out, int R, int RS, int RW = 0) {
    if (RW == 0) RW = RS;
    const size_t lds = 8 * 8 * 1024;
    (void)hipFuncSetAttribute((const void*)k<SEG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const size_t region = (size_t)R * RS;
    int passes = (int)(((size_t)48 << 20) / ((size_t)R * RW));
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<SEG>), dim3(256), dim3(512), lds, 0, buf, R, RS, RW, 2, out);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k<SEG>), dim3(256), dim3(512), lds, 0, buf, R, RS, RW, passes, out);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    const double bytes = 256.0 * 8 * (double)(passes * ((R / (64 / (SEG / 16))) * (RW / SEG) / 8) / 8 * 8) * 1024.0;
    printf("row segments of %3d B (%2d rows per 1-KiB piece), [%d rows %d B apart] per XCD: %7.1f us  %6.2f TB/s  %6.2f G requests of %d B per s and XCD (%s)\n", SEG,
           64 / (SEG / 16), R, RS, best * 1e3, bytes / (best * 1e-3) / 1e12, bytes / SEG / (best * 1e-3) / 8 / 1e9, SEG, hipGetErrorString(hipGetLastError()));
}

int main() {
    unsigned char* buf; float* out;
    (void)hipMalloc(&buf, (size_t)160 << 20);
    (void)hipMalloc(&out, 65536 * 4);
    (void)hipMemset(buf, 1, (size_t)160 << 20);
    for (int RS : {2048, 8192, 20096}) {
        const int R = RS == 20096 ? 96 : (2 << 20) / RS;           // ~2 MB per XCD
        run<64>(buf, out, R / 16 * 16, RS);
        run<128>(buf, out, R / 16 * 16, RS);
        run<256>(buf, out, R / 16 * 16, RS);
    }
    // one row per 4-KiB page (row stride 4096 + 64 / 8192 + 64 B): does the number of PAGES a CU walks matter?  (the GEMM's tile is 416 rows
    // 20 KB apart); only the first 512 B of every row are streamed so that the footprint stays L2 resident whatever the row count
    printf("== rows one page apart, 512 B of each row streamed (L2 resident), 64-B and 128-B segments ==\n");
    for (int R : {96, 416, 1024, 2048}) {
        run<64>(buf, out, R / 16 * 16, 4160, 512);
        run<128>(buf, out, R / 16 * 16, 4160, 512);
    }
    return 0;
}
