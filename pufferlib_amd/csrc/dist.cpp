// dist.cpp — RCCL (librccl.so.1, the "nccl" of ROCm) bound at run time with dlopen, so the library has no link-time
// dependency on it and a box without RCCL still loads everything else.  One communicator per process (one process per
// GPU); collectives are enqueued on the caller's HIP stream, i.e. in order with the kernels around them, without the
// stream hand-off events a framework-side collective needs.  There is no reference counterpart.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>

#include "common.hpp"

namespace pfa {
namespace {

// Minimal RCCL surface (include/rccl/rccl.h): opaque handles and the enum values we use.
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { kNcclSuccess = 0 };
enum { kNcclFloat32 = 7, kNcclFloat64 = 8 };  // ncclDataType_t
enum { kNcclSum = 0 };                         // ncclRedOp_t

struct Api {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommCount)(ncclComm_t, int *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Api g_api;
ncclComm_t g_comm = nullptr;
int g_rank = 0, g_world = 1;
unsigned long long g_rccl_calls = 0;

int load_api() {
    if (g_api.handle) return 0;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    PFA_REQUIRE(h != nullptr, "dist: cannot dlopen librccl.so.1: %s", dlerror());
    g_api.GetUniqueId = (decltype(g_api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_api.CommInitRank = (decltype(g_api.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_api.CommDestroy = (decltype(g_api.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_api.AllReduce = (decltype(g_api.AllReduce))dlsym(h, "ncclAllReduce");
    g_api.GetErrorString = (decltype(g_api.GetErrorString))dlsym(h, "ncclGetErrorString");
    g_api.CommCount = (decltype(g_api.CommCount))dlsym(h, "ncclCommCount");
    PFA_REQUIRE(g_api.GetUniqueId && g_api.CommInitRank && g_api.CommDestroy && g_api.AllReduce,
                "dist: librccl is missing a required symbol");
    g_api.handle = h;
    return 0;
}

#define PFA_CHECK_NCCL(expr)                                                                          \
    do {                                                                                              \
        int _r = (expr);                                                                              \
        if (_r != kNcclSuccess) {                                                                     \
            set_error("%s:%d %s -> rccl error %d (%s)", __FILE__, __LINE__, #expr, _r,                \
                      g_api.GetErrorString ? g_api.GetErrorString(_r) : "?");                         \
            return -3;                                                                                \
        }                                                                                             \
    } while (0)

}  // namespace

// csrc/p2p.hip: one-shot all-reduce over peer-mapped memory for buckets that fit its slots
bool p2p_ready();
int p2p_world();
bool p2p_fits(size_t bytes);
int p2p_all_reduce(void *buf, size_t count, bool f64, hipStream_t stream);
unsigned long long p2p_calls();
size_t p2p_capacity();

int dist_world() { return g_comm ? g_world : (p2p_ready() ? p2p_world() : 1); }
bool dist_ready() { return g_comm != nullptr || p2p_ready(); }

int dist_all_reduce(void *buf, size_t count, bool f64, hipStream_t stream) {
    if (p2p_fits(count * (f64 ? 8 : 4))) return p2p_all_reduce(buf, count, f64, stream);
    PFA_REQUIRE(g_comm != nullptr, "dist: communicator not initialised");
    ScopedKernelTimer timer("rccl_all_reduce", stream);   // (bench.py's per-collective figure: whatever RCCL enqueues for this call)
    PFA_CHECK_NCCL(g_api.AllReduce(buf, buf, count, f64 ? kNcclFloat64 : kNcclFloat32, kNcclSum, g_comm, stream));
    ++g_rccl_calls;
    return 0;
}

}  // namespace pfa

using namespace pfa;

extern "C" int pfa_dist_unique_id(uint8_t *id128_host) {
    PFA_REQUIRE(id128_host != nullptr, "dist.unique_id: null buffer");
    if (int rc = load_api()) return rc;
    ncclUniqueId id;
    PFA_CHECK_NCCL(g_api.GetUniqueId(&id));
    std::memcpy(id128_host, id.internal, 128);
    return 0;
}

extern "C" int pfa_dist_init(const uint8_t *id128_host, int32_t rank, int32_t world) {
    PFA_REQUIRE(id128_host != nullptr && world >= 1 && rank >= 0 && rank < world, "dist.init: bad arguments");
    if (int rc = load_api()) return rc;
    if (g_comm) {
        PFA_CHECK_NCCL(g_api.CommDestroy(g_comm));
        g_comm = nullptr;
    }
    ncclUniqueId id;
    std::memcpy(id.internal, id128_host, 128);
    PFA_CHECK_NCCL(g_api.CommInitRank(&g_comm, world, id, rank));
    g_rank = rank;
    g_world = world;
    return 0;
}

extern "C" int pfa_dist_finalize(void) {
    if (g_comm) {
        PFA_CHECK_NCCL(g_api.CommDestroy(g_comm));
        g_comm = nullptr;
    }
    g_world = 1;
    g_rank = 0;
    return 0;
}

// What is actually up, for the bench line and the logs: out[0] RCCL communicator present, [1] its ncclCommCount (0 if none),
// [2] peer path open, [3] its world size, [4] its slot capacity in bytes, [5] all-reduces that went over the peer path,
// [6] all-reduces that went over RCCL, [7] pfa_p2p_status (0 ok, 1 a wait ran out, -1 closed).
extern "C" int pfa_dist_info(int64_t *out8) {
    PFA_REQUIRE(out8 != nullptr, "dist.info: null buffer");
    int n = 0;
    if (g_comm && g_api.CommCount && g_api.CommCount(g_comm, &n) != kNcclSuccess) n = -1;
    out8[0] = g_comm != nullptr;
    out8[1] = g_comm ? n : 0;
    out8[2] = p2p_ready();
    out8[3] = p2p_ready() ? p2p_world() : 0;
    out8[4] = (int64_t)p2p_capacity();
    out8[5] = (int64_t)p2p_calls();
    out8[6] = (int64_t)g_rccl_calls;
    out8[7] = pfa_p2p_status();
    return 0;
}

extern "C" int pfa_dist_all_reduce_f32(float *buf, int64_t count, pfa_stream_t stream) {
    PFA_REQUIRE(buf && count >= 0, "dist.all_reduce: bad arguments");
    return dist_all_reduce(buf, (size_t)count, false, (hipStream_t)stream);
}

extern "C" int pfa_dist_all_reduce_f64(double *buf, int64_t count, pfa_stream_t stream) {
    PFA_REQUIRE(buf && count >= 0, "dist.all_reduce: bad arguments");
    return dist_all_reduce(buf, (size_t)count, true, (hipStream_t)stream);
}
