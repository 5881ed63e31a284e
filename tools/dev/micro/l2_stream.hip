// L2 read bandwidth of an L2-RESIDENT footprint, all 256 CUs streaming.  Every workgroup (= one CU; blockIdx & 7 =
// its XCD) streams the REGION of its XCD (<= 3 MB, so the eight regions sit in the eight 4-MiB L2s) `passes` times, as 1-KiB pieces
// (64 lanes x 16 B), U pieces in flight per loader wave, LW loader waves per CU:
//   MODE 0  global_load_dwordx4 -> VGPR (xor-folded)
//   MODE 1  buffer_load_dwordx4 ... lds (LDS-DMA, no VGPRs), counted vmcnt
//   MODE 2  global_load_dwordx4 -> VGPR -> ds_write_b128
// MW partner waves (one per SIMD when MW = 4) issue v_mfma_f32_16x16x32_bf16 back to back until the loaders are done.
// Reports TB/s over the whole chip (hipEvents) next to MI355X_MICROARCH.md's 34.5 TB/s.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int U>
__global__ __launch_bounds__(1024) void k(const uint4* __restrict__ buf, unsigned region_bytes, int passes, int LW, int spread, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_done;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) s_done = 0;
    __syncthreads();
    if (wave >= LW) {                                       // partner waves: the matrix pipe busy beside the stream
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)1.0f; }
        while (__hip_atomic_load(&s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < LW) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
        }
        out[(blockIdx.x * 1024 + tid) & 0xffff] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
        return;
    }
    // region of this CU's XCD (spread = 0: every CU its own slice of one chip-wide footprint instead -- the cross-XCD case)
    const unsigned xcd = blockIdx.x & 7, cu = blockIdx.x >> 3;
    const unsigned char* base = reinterpret_cast<const unsigned char*>(buf) + (spread ? (size_t)xcd * region_bytes : 0);
    const unsigned npieces = region_bytes >> 10;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, (int)region_bytes, 0x00027000);
    unsigned p = (cu * (npieces / 32) + wave) % npieces;    // CUs of an XCD start at different places of the region
    const unsigned step = LW;
    const unsigned total = (unsigned)passes * (npieces / LW);
    uint4 fold = {0, 0, 0, 0};
    if (MODE == 1) {
        unsigned char* slot = smem + (size_t)wave * U * 1024;
        // two half groups: U / 2 .. U pieces in flight
        for (unsigned i = 0; i < total; i += U) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int u = 0; u < U / 2; ++u) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(uintptr_t)(slot + (h * (U / 2) + u) * 1024),
                                                             16, p * 1024 + lane * 16, 0, 0, 0);
                    p += step;
                    p = p >= npieces ? p - npieces : p;
                }
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U / 2) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        fold = *reinterpret_cast<const uint4*>(slot + lane * 16);
    } else {
        uint4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            r[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, p * 1024 + lane * 16, 0, 0));
            p += step;
            p = p >= npieces ? p - npieces : p;
        }
        unsigned char* slot = smem + (size_t)wave * U * 1024 + lane * 16;
        for (unsigned i = U; i < total; i += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (MODE == 2) *reinterpret_cast<uint4*>(slot + u * 1024) = r[u];
                else { fold.x ^= r[u].x; fold.y ^= r[u].y; fold.z ^= r[u].z; fold.w ^= r[u].w; }
                r[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, p * 1024 + lane * 16, 0, 0));
                p += step;
                p = p >= npieces ? p - npieces : p;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { fold.x ^= r[u].x; fold.y ^= r[u].y; }
        if (MODE == 2) fold.z ^= *reinterpret_cast<const unsigned*>(slot);
    }
    out[(blockIdx.x * 1024 + tid) & 0xffff] = (float)(fold.x ^ fold.y ^ fold.z ^ fold.w);
    if (lane == 0) __hip_atomic_fetch_add(&s_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int MODE, int U>
void run(const uint4* buf, float* out, unsigned region, int LW, int MW, int spread, int grid = 256) {
    const size_t lds = (size_t)LW * U * 1024;
    (void)hipFuncSetAttribute((const void*)k<MODE, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const unsigned npieces = region >> 10;
    int passes = (int)(((size_t)48 << 20) / region);        // ~48 MB per CU
    if (passes < 2) passes = 2;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, U>), dim3(grid), dim3((LW + MW) * 64), lds, 0, buf, region, 2, LW, spread, out);     // warm the L2s
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k<MODE, U>), dim3(grid), dim3((LW + MW) * 64), lds, 0, buf, region, passes, LW, spread, out);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    const double bytes = (double)grid * passes * (npieces / LW) / U * U * LW * 1024.0;
    const char* names[] = {"global_load->VGPR", "buffer_load...lds", "load->ds_write_b128"};
    printf("%-20s region %4u KB/%s  loaders %2d x %d KB in flight = %3d KB/CU  mfma waves %d  grid %3d: %7.1f us  %6.2f TB/s  (%5.1f GB/s per CU) %s\n",
           names[MODE], region >> 10, spread ? "XCD" : "chip", LW, U, LW * U, MW, grid, best * 1e3, bytes / (best * 1e-3) / 1e12,
           bytes / grid / (best * 1e-3) / 1e9, hipGetErrorString(hipGetLastError()));
}

template <int MODE>
void sweep(const uint4* buf, float* out, unsigned region, int spread) {
    run<MODE, 4>(buf, out, region, 4, 0, spread);           // 16 KB / CU, 1 wave / SIMD
    run<MODE, 8>(buf, out, region, 4, 0, spread);           // 32
    run<MODE, 8>(buf, out, region, 8, 0, spread);           // 64, 2 waves / SIMD
    run<MODE, 8>(buf, out, region, 16, 0, spread);          // 128, 4 waves / SIMD
    run<MODE, 16>(buf, out, region, 8, 0, spread);          // 128, 2 waves / SIMD
    run<MODE, 8>(buf, out, region, 4, 4, spread);           // 32 KB + a partner MFMA wave per SIMD
    run<MODE, 8>(buf, out, region, 8, 4, spread);           // 64 KB + partner
    run<MODE, 16>(buf, out, region, 8, 4, spread);          // 128 KB + partner
}

int main(int argc, char** argv) {
    const unsigned region = (argc > 1 ? atoi(argv[1]) : 2048) << 10;       // KB per XCD
    uint4* buf; float* out;
    const size_t total = (size_t)8 * region > ((size_t)512 << 20) ? (size_t)8 * region : ((size_t)512 << 20);
    (void)hipMalloc(&buf, total);
    (void)hipMalloc(&out, 65536 * 4);
    (void)hipMemset(buf, 1, total);
    printf("== per-XCD regions of %u KB (L2 resident: 8 x %u KB), all 256 CUs; guide: L2 ~34.5 TB/s, HBM ~6.3 achievable ==\n", region >> 10, region >> 10);
    sweep<0>(buf, out, region, 1);
    sweep<1>(buf, out, region, 1);
    sweep<2>(buf, out, region, 1);
    printf("== one CU alone (grid 1) and one XCD alone (grid 8 -> one CU per XCD; 32 CUs of one XCD cannot be selected) ==\n");
    run<0, 8>(buf, out, region, 8, 0, 1, 1);
    run<1, 8>(buf, out, region, 8, 0, 1, 1);
    run<0, 8>(buf, out, region, 8, 0, 1, 8);
    run<0, 8>(buf, out, region, 8, 0, 1, 64);
    run<0, 8>(buf, out, region, 8, 0, 1, 128);
    printf("== the same footprint read by EVERY CU whatever its XCD (one 1.3-MB weight image shared chip wide: each L2 holds a copy) ==\n");
    run<0, 8>(buf, out, 1344 << 10, 8, 0, 0);
    run<1, 8>(buf, out, 1344 << 10, 8, 0, 0);
    run<1, 8>(buf, out, 1344 << 10, 8, 4, 0);
    printf("== 32-KB region (fits the CU's 32-KB L1 / TCP): the L1 hit rate ==\n");
    run<0, 8>(buf, out, 16 << 10, 8, 0, 1);
    run<1, 8>(buf, out, 16 << 10, 8, 0, 1);
    printf("== 64 MB per XCD region (HBM / Infinity Cache stream) ==\n");
    run<0, 8>(buf, out, 64u << 20, 8, 0, 1);
    run<1, 8>(buf, out, 64u << 20, 8, 0, 1);
    return 0;
}
