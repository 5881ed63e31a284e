// The forward's ONE collective behind the C ABI: an all-gather of the [rows_local, num_labels] logits over RCCL (xGMI inside a
// node).  The reference is single-GPU (hard-coded cuda:0, Multi_GCN_Multihead_att.py:85,465,493; its DataParallel line is
// commented out, engine/Multi_GCN_Multihead_Att_engine.py:365); its eval forward has no cross-sample reduction, so sharding
// the batch over ranks and gathering the logits is exact.
//
// RCCL is bound at RUN time (dlopen), not at link time: a Python process that imported torch already carries torch's own
// librccl.so.1, and two RCCL instances in one process do not share device state.  The copy that is already loaded is reused
// (RTLD_NOLOAD); a plain C/C++ host gets the system library.  Nothing here allocates device memory or synchronises; the
// all-gather is enqueued on the caller's stream and captures into a hipGraph like any RCCL collective.
#include "common.hpp"
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = "";
};

extern const char* g_rccl_why;
Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // the instance this process already uses
        for (const char* n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.handle) r.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!r.handle) {
            snprintf(r.why, sizeof(r.why), "librccl.so.1 not loadable: %s", dlerror());
            g_rccl_why = r.why;
            return;
        }
        bool ok = true;
        auto sym = [&](const char* name) {
            void* p = dlsym(r.handle, name);
            if (!p) {
                ok = false;
                snprintf(r.why, sizeof(r.why), "librccl: symbol %s missing", name);
            }
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        if (!ok) r.handle = nullptr;
        if (!r.handle) g_rccl_why = r.why;
    });
    return r.handle ? &r : nullptr;
}

const char* g_rccl_why = "RCCL unavailable (librccl.so.1 could not be bound)";
const char* rccl_why() { return g_rccl_why; }

#define MG_RCCL(call, what)                                                                   \
    do {                                                                                      \
        const ncclResult_t r_ = (call);                                                       \
        if (r_ != ncclSuccess) {                                                              \
            mgnns_set_error("%s: %s", what, R->GetErrorString ? R->GetErrorString(r_) : "RCCL error"); \
            return MGNNS_ERR_LAUNCH;                                                          \
        }                                                                                     \
    } while (0)

}  // namespace

extern "C" int mgnns_comm_unique_id(void* id, size_t bytes) {
    MG_REQUIRE(id && bytes >= NCCL_UNIQUE_ID_BYTES, "mgnns_comm_unique_id: need a %d-byte buffer", NCCL_UNIQUE_ID_BYTES);
    Rccl* R = rccl();
    if (!R) { mgnns_set_error("mgnns_comm_unique_id: %s", rccl_why()); return MGNNS_ERR_UNSUPP; }
    ncclUniqueId u;
    MG_RCCL(R->GetUniqueId(&u), "ncclGetUniqueId");
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

extern "C" int mgnns_comm_init_rank(int world, int rank, const void* id, size_t bytes, mgnns_comm_t* comm) {
    MG_REQUIRE(comm && id && bytes >= NCCL_UNIQUE_ID_BYTES, "mgnns_comm_init_rank: null pointer / short id");
    MG_REQUIRE(world >= 1 && rank >= 0 && rank < world, "mgnns_comm_init_rank: rank %d of %d", rank, world);
    Rccl* R = rccl();
    if (!R) { mgnns_set_error("mgnns_comm_init_rank: %s", rccl_why()); return MGNNS_ERR_UNSUPP; }
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    MG_RCCL(R->CommInitRank(&c, world, u, rank), "ncclCommInitRank");      // binds to the calling thread's current device
    *comm = reinterpret_cast<mgnns_comm_t>(c);
    return 0;
}

extern "C" int mgnns_comm_init_all(int ndev, const int* devices, mgnns_comm_t* comms) {
    MG_REQUIRE(comms && ndev >= 1, "mgnns_comm_init_all: ndev=%d", ndev);
    Rccl* R = rccl();
    if (!R) { mgnns_set_error("mgnns_comm_init_all: %s", rccl_why()); return MGNNS_ERR_UNSUPP; }
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have < ndev) {
        mgnns_set_error("mgnns_comm_init_all: %d devices requested, %d visible", ndev, have);
        return MGNNS_ERR_ARG;
    }
    MG_RCCL(R->CommInitAll(reinterpret_cast<ncclComm_t*>(comms), ndev, devices), "ncclCommInitAll");
    return 0;
}

extern "C" int mgnns_comm_info(mgnns_comm_t comm, int* world, int* rank) {
    MG_REQUIRE(comm && world && rank, "mgnns_comm_info: null pointer");
    Rccl* R = rccl();
    if (!R) { mgnns_set_error("mgnns_comm_info: %s", rccl_why()); return MGNNS_ERR_UNSUPP; }
    MG_RCCL(R->CommCount(reinterpret_cast<ncclComm_t>(comm), world), "ncclCommCount");
    MG_RCCL(R->CommUserRank(reinterpret_cast<ncclComm_t>(comm), rank), "ncclCommUserRank");
    return 0;
}

extern "C" int mgnns_allgather_logits(mgnns_comm_t comm, const float* local, int rows_local, int num_labels, float* all,
                                      mgnns_stream_t stream) {
    MG_REQUIRE(comm, "mgnns_allgather_logits: null communicator");
    MG_REQUIRE(rows_local >= 0 && num_labels > 0, "mgnns_allgather_logits: rows=%d labels=%d", rows_local, num_labels);
    if (rows_local == 0) return 0;
    MG_REQUIRE(local && all, "mgnns_allgather_logits: null pointer");
    Rccl* R = rccl();
    if (!R) { mgnns_set_error("mgnns_allgather_logits: %s", rccl_why()); return MGNNS_ERR_UNSUPP; }
    MG_RCCL(R->AllGather(local, all, (size_t)rows_local * num_labels, ncclFloat32, reinterpret_cast<ncclComm_t>(comm),
                         (hipStream_t)stream),
            "ncclAllGather");
    return 0;
}

// One thread driving several communicators (mgnns_comm_init_all) brackets its per-device all-gathers with these, as RCCL
// requires for multi-device calls from a single thread; one process per GPU does not need them.
extern "C" int mgnns_comm_group_start(void) {
    Rccl* R = rccl();
    if (!R) { mgnns_set_error("mgnns_comm_group_start: %s", rccl_why()); return MGNNS_ERR_UNSUPP; }
    MG_RCCL(R->GroupStart(), "ncclGroupStart");
    return 0;
}
extern "C" int mgnns_comm_group_end(void) {
    Rccl* R = rccl();
    if (!R) { mgnns_set_error("mgnns_comm_group_end: %s", rccl_why()); return MGNNS_ERR_UNSUPP; }
    MG_RCCL(R->GroupEnd(), "ncclGroupEnd");
    return 0;
}

extern "C" int mgnns_comm_destroy(mgnns_comm_t comm) {
    if (!comm) return 0;
    Rccl* R = rccl();
    if (!R) { mgnns_set_error("mgnns_comm_destroy: %s", rccl_why()); return MGNNS_ERR_UNSUPP; }
    MG_RCCL(R->CommDestroy(reinterpret_cast<ncclComm_t>(comm)), "ncclCommDestroy");
    return 0;
}
