// Error channel + ABI version of libmgnns_hip.so.
#include "common.hpp"

static thread_local char g_err[512] = "";

void mgnns_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mgnns_last_error(void) { return g_err; }
extern "C" int mgnns_abi_version(void) { return 3; }
