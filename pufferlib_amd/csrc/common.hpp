// common.hpp — error plumbing and small device helpers shared by every kernel file (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "../../include/pufferlib_amd.h"

namespace pfa {

void set_error(const char *fmt, ...);
bool timing_enabled();
void *timing_begin(const char *name, hipStream_t stream);  // records the start event when timing is on
void timing_end(void *stop, hipStream_t stream);
bool timing_ext_mode();                                                  // launches that carry their own events (hipExtLaunchKernelGGL)
bool timing_pair(const char *name, hipEvent_t *start, hipEvent_t *stop);  // for launches that carry their own events (hip_ext.h)

struct ScopedKernelTimer {  // brackets ONE kernel launch with HIP events on its own stream (bench.py roofline)
    void *stop;
    hipStream_t stream;
    ScopedKernelTimer(const char *name, hipStream_t s) : stop(timing_begin(name, s)), stream(s) {}
    ~ScopedKernelTimer() { timing_end(stop, stream); }
};

#define PFA_CHECK_HIP(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::pfa::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return -1;                                                                        \
        }                                                                                     \
    } while (0)

#define PFA_REQUIRE(cond, ...)              \
    do {                                    \
        if (!(cond)) {                      \
            ::pfa::set_error(__VA_ARGS__);  \
            return -2;                      \
        }                                   \
    } while (0)

#define PFA_LAUNCH_CHECK() PFA_CHECK_HIP(hipGetLastError())

// csrc/dist.cpp: process-wide RCCL communicator (one process per GPU)
int dist_world();
bool dist_ready();
int dist_all_reduce(void *buf, size_t count, bool f64, hipStream_t stream);

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// The wave index as a SCALAR (workgroups are one-dimensional multiples of 64 threads everywhere in this library): what is derived from
// it — roles, tile indices, LDS slice bases, validity tests — then lives on the scalar unit with scalar branches instead of VGPRs and
// exec-masked VALU code (the compiler does not treat threadIdx.x >> 6 as wave-uniform; ppo_mlp_grad_kernel: 3 spilled dwords and
// ~80 VALU instructions fewer).
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Workgroup barrier that orders LDS traffic only: `s_waitcnt lgkmcnt(0); s_barrier`, never a wait on the vector-memory counter.
// (hipcc 7.2 compiles `__syncthreads()` to the same two instructions on gfx950 in the default non-tgsplit mode — checked in the
// ISA — so this states the intent rather than working around the compiler: the kernels that use it keep experience stores in
// flight across their barriers and must not start waiting for them if a fence over every address space ever gets stricter.)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

}  // namespace pfa
