// Error channel + ABI version of libmgnns_hip.so.
#include "common.hpp"
#include <stdlib.h>
#include <mutex>
#include <set>
#include <utility>

static thread_local char g_err[512] = "";

void mgnns_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Raise a kernel's dynamic-LDS limit once per (kernel, device): the attribute lives on the device's code object, so a
// process that drives several GPUs has to set it on each of them.
int mg_ensure_dyn_lds(const void* fn, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return MGNNS_ERR_LAUNCH;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({fn, dev})) return 0;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        mgnns_set_error("hipFuncSetAttribute(dynamic LDS = %d B) failed: %s", bytes, hipGetErrorString(e));
        return MGNNS_ERR_LAUNCH;
    }
    done.insert({fn, dev});
    return 0;
}

// ---- status word of the persistent launches -----------------------------------------------------------------------
// The kernels that hand data between workgroups of one launch (label GCN item queue, channel-tail / layer-tail clusters) bound
// every wait; a wait that runs out raises this word (host-pinned, device-mapped memory registered by the host side) and the
// launch drains.  The next persistent launch of the process reports it: no hidden synchronisation, no hang.
static int32_t* g_status = nullptr;

int32_t* mg_status_word() { return g_status; }

int mg_check_status(const char* who) {
    int32_t* p = g_status;
    if (!p) return 0;
    const int32_t code = __atomic_load_n(p, __ATOMIC_RELAXED);
    if (code == 0) return 0;
    __atomic_store_n(p, 0, __ATOMIC_RELAXED);
    mgnns_set_error("%s: an EARLIER persistent launch gave up a bounded wait (status %d: %s) -- its results and everything computed "
                    "from them since are invalid; the device is fine, re-run the forward",
                    who, (int)code,
                    code == MGNNS_STATUS_LABEL_GCN_TIMEOUT ? "label GCN item queue"
                    : code == MGNNS_STATUS_CLUSTER_TIMEOUT ? "workgroup-cluster exchange"
                    : code == MGNNS_STATUS_BAD_PLAN      ? "a masked attention launch refused a plan of another kind or batch (it wrote nothing)"
                                                         : "unknown");
    return MGNNS_ERR_LAUNCH;
}

extern "C" int mgnns_set_status_word(int32_t* host_pinned) {
    g_status = host_pinned;
    if (host_pinned) __atomic_store_n(host_pinned, 0, __ATOMIC_RELAXED);
    return 0;
}

extern "C" int mgnns_take_status(void) {
    int32_t* p = g_status;
    if (!p) return 0;
    return (int)__atomic_exchange_n(p, 0, __ATOMIC_RELAXED);
}

// CU count of the current device (hipDeviceAttributeMultiprocessorCount), cached per device; 0 on failure (error text set)
int mg_cu_count() {
    static std::mutex mu;
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        mgnns_set_error("cannot query the current device");
        return 0;
    }
    std::lock_guard<std::mutex> lk(mu);
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
            mgnns_set_error("cannot query the CU count of device %d", dev);
            return 0;
        }
        cached[dev] = n;
    }
    return cached[dev];
}

// integer tuning knob from the environment, read ONCE per process (the launchers sit on the per-forward path)
int mg_env_int(const char* name, int fallback, int slot) {
    static std::mutex mu;
    static bool have[16] = {false};
    static int val[16] = {0};
    std::lock_guard<std::mutex> lk(mu);
    if (!have[slot]) {
        const char* e = getenv(name);
        val[slot] = e ? atoi(e) : fallback;
        have[slot] = true;
    }
    return val[slot];
}

extern "C" const char* mgnns_last_error(void) { return g_err; }
extern "C" int mgnns_abi_version(void) { return 18; }

namespace {
// which XCD (hardware XCC_ID) a workgroup runs on, per block index
__global__ void xcd_probe_kernel(int* out) {
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(20, 0, 4)" : "=s"(id));      // HW_REG_XCC_ID, bits 3:0
        out[blockIdx.x] = (int)id;
    }
}
}  // namespace

// Several kernels place work by `blockIdx.x & 7 == XCD` (feature slabs of the SpMM kernels, the dense GEMM's row-block ranges,
// the tiled SpMM's slab order): HIP promises no workgroup -> XCD map, ONLY SPEED depends on it, and this is the measurement that
// says whether the assumption holds on this device / runtime: 1024 one-wave workgroups report their XCC_ID.
// out[0] = 1 if block b always ran on one XCD per value of b & 7 and the eight values map to eight different XCDs, else 0;
// out[1 + k] = the XCC_ID seen for b & 7 == k (-1: several).  Synchronises the device (call it at set-up, not in a forward).
extern "C" int mgnns_xcd_probe(int32_t* out9) {
    MG_REQUIRE(out9, "mgnns_xcd_probe: null pointer");
    constexpr int NB = 1024;
    int* d = nullptr;
    if (hipMalloc(&d, NB * sizeof(int)) != hipSuccess) {
        mgnns_set_error("mgnns_xcd_probe: hipMalloc failed");
        return MGNNS_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(xcd_probe_kernel, dim3(NB), dim3(64), 0, 0, d);
    int h[NB];
    const hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) {
        mgnns_set_error("mgnns_xcd_probe: %s", hipGetErrorString(e));
        return MGNNS_ERR_LAUNCH;
    }
    int ok = 1;
    for (int k = 0; k < 8; ++k) {
        int id = h[k];
        for (int b = k; b < NB; b += 8)
            if (h[b] != id) id = -1;
        out9[1 + k] = id;
        if (id < 0) ok = 0;
        for (int j = 0; j < k; ++j)
            if (out9[1 + j] == id) ok = 0;
    }
    out9[0] = ok;
    return 0;
}

namespace {
__global__ void stamp_kernel(unsigned long long* slots, int idx) { slots[idx] = __builtin_amdgcn_s_memrealtime(); }
__global__ void spin_kernel(unsigned long long ticks, unsigned long long* slots, int idx) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (slots) slots[idx] = __builtin_amdgcn_s_memrealtime();
}
}  // namespace

extern "C" int mgnns_debug_spin(int microseconds, uint64_t* slots, int idx, mgnns_stream_t stream) {
    MG_REQUIRE(microseconds >= 0 && microseconds <= 100000 && idx >= 0, "mgnns_debug_spin: bad arguments");
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long)microseconds * 100ull,
                       reinterpret_cast<unsigned long long*>(slots), idx);
    MG_CHECK_LAUNCH("mgnns_debug_spin");
    return 0;
}

extern "C" int mgnns_debug_stamp(uint64_t* slots, int idx, mgnns_stream_t stream) {
    MG_REQUIRE(slots && idx >= 0, "mgnns_debug_stamp: bad arguments");
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long*>(slots), idx);
    MG_CHECK_LAUNCH("mgnns_debug_stamp");
    return 0;
}
