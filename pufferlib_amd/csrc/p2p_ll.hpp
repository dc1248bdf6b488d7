// p2p_ll.hpp — flag-in-data exchange over peer-mapped memory: the device half of the data-parallel optimizer step that runs INSIDE
// the kernel that reduces the gradient partials (ppo_update.hip: ppo_reduce_adam_kernel) instead of as a launch of its own
// (no reference counterpart).
//
// Every value travels as ONE 8-byte store {float bits, sequence number}: a reader that sees the sequence number of this call
// sees the value that was stored with it (an aligned 8-byte store is a single transaction on the fabric), so the protocol needs
// neither a fence nor any ordering between different stores — nothing about how xGMI orders posted writes from different
// workgroups is assumed.  Layout in every rank's buffer: [2 phases][R sources][entries] of uint64.  A call with sequence
// number s uses phase s & 1; rank A can only start call s + 2 (same phase again) after it finished call s + 1, which needed
// every peer's push of call s + 1, which those peers issued after they had finished reading call s — two phases suffice.
// A wait is bounded in wall-clock time (100 MHz counter): when it runs out the status word is raised and the value comes back
// NaN, so a lost peer poisons parameters and losses instead of hanging the GPU or handing back a partial sum.
#pragma once
#include "common.hpp"

namespace pfa {

constexpr int kP2pMaxRanks = 8;

struct LlArgs {
    unsigned long long *base[kP2pMaxRanks];   // every rank's LL area as mapped into THIS process (base[rank] = the local one)
    int rank, world;
    unsigned entries;                         // per (phase, source) slot
    unsigned seq;                             // this call's sequence number (never 0: the area starts zeroed)
    int *status;                              // host-pinned status word (shared with the flag-based path)
    long long timeout_ticks;
    unsigned long long *wait_stats;           // device counters (pfa_p2p_wait_stats): [0] += ticks a workgroup spent waiting for peers, [1] += 1
};

__device__ __forceinline__ unsigned long long ll_pack(float v, unsigned seq) {
    return ((unsigned long long)seq << 32) | (unsigned long long)__float_as_uint(v);
}

// Publish v as entry idx of this rank's slot on every peer.
__device__ __forceinline__ void ll_push(const LlArgs &d, unsigned idx, float v) {
    const size_t off = ((size_t)(d.seq & 1u) * d.world + d.rank) * d.entries + idx;
    const unsigned long long w = ll_pack(v, d.seq);
    for (int q = 0; q < d.world; ++q)
        if (q != d.rank) __hip_atomic_store(d.base[q] + off, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Sum of entry idx over all ranks in rank order (this rank's own term from the register).  Identical bits on every rank.
// `waited` (optional) += the 100 MHz ticks this lane spent spinning for entries that had not arrived yet.
__device__ __forceinline__ float ll_wait_sum(const LlArgs &d, unsigned idx, float mine, long long *waited = nullptr) {
    float s = 0.0f;
    bool lost = false;
    for (int r = 0; r < d.world; ++r) {
        float v = mine;
        if (r != d.rank) {
            const unsigned long long *p = d.base[d.rank] + ((size_t)(d.seq & 1u) * d.world + r) * d.entries + idx;
            unsigned long long w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned)(w >> 32) != d.seq) {
                const long long t0 = (long long)wall_clock64();
                int spin = 0;
                do {
                    __builtin_amdgcn_s_sleep(1);
                    w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if ((++spin & 255) == 0 && (long long)wall_clock64() - t0 > d.timeout_ticks) {
                        lost = true;
                        break;
                    }
                } while ((unsigned)(w >> 32) != d.seq);
                if (waited) *waited += (long long)wall_clock64() - t0;
            }
            v = __uint_as_float((unsigned)w);
        }
        s = r == 0 ? v : s + v;
    }
    if (lost) {
        __hip_atomic_store(d.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        s = __builtin_nanf("");
    }
    return s;
}

// Peer-wait telemetry: the longest wait of a workgroup's exchange wave (all 64 lanes active) goes into the device counters, one
// atomic pair per workgroup.  Summed over a run, ticks / workgroups = the mean time a launch stood waiting for its slowest peer —
// transport latency when the ranks arrive together, rank skew on top of it when they do not (bench.py: dist.wait_us_per_step).
__device__ __forceinline__ void ll_wait_report(const LlArgs &d, long long waited) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const long long other = __shfl_xor(waited, o, 64);
        waited = other > waited ? other : waited;
    }
    if (d.wait_stats && (threadIdx.x & 63) == 0) {
        atomicAdd(d.wait_stats, (unsigned long long)waited);
        atomicAdd(d.wait_stats + 1, 1ull);
    }
}

// A timeout is every rank's business: the rank whose wait ran out holds NaN from then on, but a peer that was merely late sees a
// complete exchange and would train on with a diverged replica.  So every exchange also carries the ranks' status words (the LAST
// entry of the slot, one lane per launch): a rank that has a raised status pushes 1, everybody sums, and a rank that sees a non-zero
// (or unreadable) sum raises its own word to 2 ("a peer reported a lost exchange").  clean_pufferl.train() checks the word after
// every update on every rank, so the timeout of optimizer step k is an error on ALL ranks in the same train() for every k but the
// update's last step (that one is reported by the first exchange of the next update).  A lane whose own status is already raised
// does not wait again (its peers may be gone: one bounded wait per launch is enough).
// Two halves, so that a kernel can put the slow parts where they cost nothing: the status word lives in host-pinned memory (a read
// over PCIe, ~1.5 us) — ll_status_push reads and publishes it at the START of the launch, under the partial sums; ll_status_wait
// collects the peers' words late, next to another wait.
__device__ __forceinline__ int ll_status_push(const LlArgs &d) {
    const int mine = __hip_atomic_load(d.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    ll_push(d, d.entries - 1u, mine ? 1.0f : 0.0f);
    return mine;
}
__device__ __forceinline__ void ll_status_wait(const LlArgs &d, int mine) {
    if (mine) return;
    const float tot = ll_wait_sum(d, d.entries - 1u, 0.0f);        // (raises the word to 1 itself when the wait runs out)
    if (tot > 0.0f) __hip_atomic_store(d.status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void ll_status_exchange(const LlArgs &d) { ll_status_wait(d, ll_status_push(d)); }

// csrc/p2p.hip
bool p2p_ll_ready(size_t entries_needed);
LlArgs p2p_ll_next();   // arguments of the next call: bumps the sequence number

}  // namespace pfa
