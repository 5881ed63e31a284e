// Helpers of the unit-queue attention cores (sq_mha_split_bf16.hip): compile-time loops, the packed-weight stream through a
// buffer resource, LDS hand-over counters, the transposing four-value row reduction.  (sq_mha_bf16.hip / sq_mha32_bf16.hip carry
// their own copies of these inside their anonymous namespaces: they are tuned against one compiler's register allocation and
// are left untouched.)
#pragma once
#include "common.hpp"

namespace mg_mha {

typedef int i32x4 __attribute__((ext_vector_type(4)));

// compile-time loop: f(IC<0>{}), f(IC<1>{}), ... -- the index is a constant expression inside f (immediate offsets / counts of
// inline-asm instructions need one)
template <int N> struct IC { static constexpr int v = N; };
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(IC<I>{});
        static_for<I + 1, N>(f);
    }
}

// Packed weights through a buffer resource: one VGPR (lane * 16) addresses every fragment, the fragment is selected by a
// wave-uniform byte offset in an SGPR.
struct WStream {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;                                   // lane * 16
};
__device__ __forceinline__ uint4 wfrag(const WStream& w, int soff) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(w.rsrc, w.voff, soff, 0));
}

// Sum over the four 16-lane rows of the wave for FOUR values at once (a transposing reduction): on return the rows of the
// result hold the row sums of [a, c, b, d] -- row 0: a, row 1: c, row 2: b, row 3: d.  v_permlane32_swap exchanges the upper
// half of its first operand with the lower half of its second, v_permlane16_swap the odd rows of the first with the even rows
// of the second.  Inline asm: both registers of a swap are read AND written; the s_nop 1 on either side cover the VALU-write ->
// swap-read and swap-write -> VALU-read hazards, which the compiler's hazard recogniser does not see through an asm block.
__device__ __forceinline__ float rows4_sum4(float a, float b, float c, float d) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    float ab = a + b, cd = c + d;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(ab), "+v"(cd));
    return ab + cd;
}

// Cross-wave hand-over inside a workgroup through LDS counters (no s_barrier).  LDS operations of a wave complete in order:
// lgkmcnt(0) in front of an arrival publishes this wave's LDS writes to whoever sees the count.
__device__ __forceinline__ int lds_arrive(int* ctr, int lane) {          // -> the count before this arrival (wave-uniform)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __builtin_amdgcn_readfirstlane(old);
}
__device__ __forceinline__ void lds_wait_ge(int* ctr, int target) {
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
        __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

}  // namespace mg_mha
