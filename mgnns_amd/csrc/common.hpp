// Shared host/device helpers of libmgnns_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/mgnns_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

void mgnns_set_error(const char* fmt, ...);
int mg_ensure_dyn_lds(const void* fn, int bytes);   // api.hip; 0 or MGNNS_ERR_LAUNCH (error text set)
int32_t* mg_status_word();                           // api.hip: the registered status word (host-pinned) or nullptr
int mg_cu_count();                                     // api.hip: CUs of the current device (cached); 0 + error text on failure
int mg_env_int(const char* name, int fallback, int slot);   // api.hip: environment knob read once per process (slot 0..15 = its cache entry)
int mg_check_status(const char* who);                // api.hip: MGNNS_ERR_LAUNCH (+ text, word cleared) if an earlier bounded wait ran out
#define MG_DYN_LDS(fn, bytes)                                                              \
    do {                                                                                   \
        if (int r_ = mg_ensure_dyn_lds(reinterpret_cast<const void*>(fn), (int)(bytes))) return r_; \
    } while (0)

#define MG_REQUIRE(cond, ...)                         \
    do {                                              \
        if (!(cond)) {                                \
            mgnns_set_error(__VA_ARGS__);             \
            return MGNNS_ERR_ARG;                     \
        }                                             \
    } while (0)

#define MG_CHECK_LAUNCH(name)                                                   \
    do {                                                                        \
        hipError_t e_ = hipGetLastError();                                      \
        if (e_ != hipSuccess) {                                                 \
            mgnns_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return MGNNS_ERR_LAUNCH;                                            \
        }                                                                       \
    } while (0)

static inline bool mg_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float mg_act(float v, int act) {
    if (act == MGNNS_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == MGNNS_ACT_LRELU2) return v > 0.0f ? v : 0.2f * v;
    return v;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence, which the compiler
// lowers to s_waitcnt vmcnt(0) lgkmcnt(0): every global load in flight -- including a prefetch requested on purpose
// for a LATER iteration -- has to land before the barrier.  Use this one where the barrier only publishes LDS writes /
// retires LDS reads and global loads are meant to stay in flight across it.
__device__ __forceinline__ void mg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 16-byte LDS read the compiler does not see as an LDS access.  hipcc puts s_waitcnt vmcnt(0) in front of every ds_read
// it knows about while an LDS-DMA (global_load_lds) is outstanding -- it cannot tell that the ring stage being read is not
// the one being filled -- which serialises a multi-stage DMA ring into "request a slice, wait for it".  Reads issued
// through this helper carry no such wait; the CALLER orders them: s_waitcnt lgkmcnt(n) through mg_lds_wait<n>(regs...)
// before the first use (LDS operations of a wave complete in order).
template <int OFF>
__device__ __forceinline__ u32x4 mg_lds_read128(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ unsigned mg_lds_addr(const void* p) { return (unsigned)(uintptr_t)p; }   // generic -> LDS offset
// s_waitcnt lgkmcnt(N) for fragments read with mg_lds_read128.  Nothing ties the consumer MFMAs to the wait (tying the fragment
// or accumulator registers as in/out asm operands made hipcc copy them around the wait -- some copies BEFORE it, i.e. before the
// data had landed): follow it with __builtin_amdgcn_sched_barrier(0), which no instruction is scheduled across.
template <int N>
__device__ __forceinline__ void mg_lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

// ---- DPP (pure VALU, no LDS crossbar) reductions -----------------------------------------------------------
// 16-lane row reductions: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror -> all 16 lanes
#define MG_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ float row16_sum(float v) {
    v += MG_DPP(v, 0xB1);
    v += MG_DPP(v, 0x4E);
    v += MG_DPP(v, 0x141);
    v += MG_DPP(v, 0x140);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, MG_DPP(v, 0xB1));
    v = fmaxf(v, MG_DPP(v, 0x4E));
    v = fmaxf(v, MG_DPP(v, 0x141));
    v = fmaxf(v, MG_DPP(v, 0x140));
    return v;
}
// full-wave max, result wave-uniform: DPP inside the four rows, then the four row results via v_readlane
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = row16_max(v);
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = row16_sum(v);
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (a + b) + (c + d);
}

// 64-lane butterfly reductions (wave = 64 on gfx950)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
