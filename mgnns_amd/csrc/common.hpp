// Shared host/device helpers of libmgnns_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/mgnns_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void mgnns_set_error(const char* fmt, ...);

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
