// p2p.hip — one-shot all-reduce for small buckets over peer-mapped device memory (no reference counterpart).
//
// The data-parallel update all-reduces one flat bucket per optimizer step — 38 KB for the MLP policy, 0.6 MB with the LSTM —
// 16 times per PPO update.  At that size a ring all-reduce is pure latency: 2 (R-1) = 14 dependent hops on 8 GPUs, each bound
// by one xGMI link.  The MI355X node is a full mesh (7 links per GPU), so every rank can instead WRITE its bucket straight
// into a slot of every peer's memory over all 7 links at once, raise a flag there, wait for the 7 flags raised in its own
// memory and add the R slots up locally, in rank order — one hop, and bit-identical sums on every rank (same values, same
// order), which keeps the replicas' parameters identical without a broadcast.
//
// Memory: each rank owns one fine-grained device allocation [2 phases][R sources][capacity] + flags [2][R][kP2pChunks]
// (uint64 sequence numbers), exported with hipIpcGetMemHandle and mapped by every peer.  A call splits the bucket over up to
// kP2pChunks workgroups; workgroup w of rank r pushes chunk w into slot (phase, r) of every peer, fences, sets flag
// (phase, r, w) = seq on every peer, waits for flags (phase, *, w) >= seq in its own memory and reduces chunk w.  Two phases
// suffice: a rank can start call s+1 (other phase) while a slow peer still reads phase s, but not call s+2 before that peer
// has pushed call s+1, i.e. finished reading call s.
//
// A lost peer ends in an error, not a hung GPU and not a silently wrong sum: the flag wait is bounded in wall-clock time
// (100 MHz counter; default 30 s, PFA_WAIT_TIMEOUT_MS), and a wait that runs out (a) raises the status word — host-pinned
// memory, so the host reads it without a copy or a synchronisation (pfa_p2p_status; clean_pufferl.train() checks it after
// every update and raises) — and (b) POISONS the chunk it could not complete with NaN, so that losses, weights and every
// isfinite check downstream fail even where nobody looks at the status.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "common.hpp"
#include "p2p_ll.hpp"

namespace pfa {

constexpr int kP2pChunks = 64;
constexpr int kP2pThreads = 512;

struct P2pPeers {
    char *base[kP2pMaxRanks];
};
struct P2pState {
    bool ready = false;
    int rank = 0, world = 1;
    size_t cap_bytes = 0;   // per (phase, source) slot
    char *local = nullptr;
    P2pPeers peers{};
    unsigned long long seq = 0;
    int *status = nullptr;         // host-pinned word (device-visible): set non-zero when a wait ran out
    long long timeout_ticks = 0;   // of the 100 MHz wall clock
    unsigned long long calls = 0;
    // flag-in-data area (p2p_ll.hpp) behind the slots and flags: [2][world][ll_entries] uint64
    size_t ll_offset = 0;
    unsigned ll_entries = 0, ll_seq = 0;
    unsigned long long ll_calls = 0;
    // peer-wait telemetry (device): [0] ticks / [1] workgroups of the flag-in-data exchanges, [2] / [3] of the flag-based all-reduces
    unsigned long long *wait_stats = nullptr;
};
static P2pState g_p2p;

__host__ __device__ inline size_t p2p_slot_offset(size_t cap, int world, int phase, int src) { return ((size_t)phase * world + src) * cap; }
__host__ __device__ inline size_t p2p_flags_offset(size_t cap, int world) { return (size_t)2 * world * cap; }
static size_t p2p_ll_offset(size_t cap, int world) {
    return align_up(p2p_flags_offset(cap, world) + (size_t)2 * world * kP2pChunks * sizeof(unsigned long long) + 256, 256);
}
// entries per (phase, source): every float of a bucket that fits a slot + the head room the gradient's fragment-order layout
// needs over the flat one (ppo_update.hip NativeLayout: up to 2048 + 24 entries more than the flat vector)
static unsigned p2p_ll_entries(size_t cap) { return (unsigned)(cap / 4 + 2304); }
static size_t p2p_total_bytes(size_t cap, int world) {
    return p2p_ll_offset(cap, world) + (size_t)2 * world * p2p_ll_entries(cap) * sizeof(unsigned long long) + 256;
}

template <typename T>
__global__ void __launch_bounds__(kP2pThreads) p2p_all_reduce_kernel(P2pPeers peers, int rank, int world, size_t cap, int phase,
                                                                    unsigned long long seq, T *buf, long long n, int *status,
                                                                    long long timeout_ticks, unsigned long long *wait_stats) {
    __shared__ int timed_out;
    __shared__ unsigned long long wait_max;
    if (threadIdx.x == 0) {
        timed_out = 0;
        wait_max = 0;
    }
    const int w = blockIdx.x;
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)w * per, hi = lo + per < n ? lo + per : n;
    // push this rank's chunk into its slot on every peer (own memory included: the reduction reads all R slots from one place)
    for (int q = 0; q < world; ++q) {
        T *dst = reinterpret_cast<T *>(peers.base[q] + p2p_slot_offset(cap, world, phase, rank));
        for (long long i = lo + threadIdx.x; i < hi; i += kP2pThreads) __builtin_nontemporal_store(buf[i], dst + i);
    }
    __threadfence_system();
    __syncthreads();
    if ((int)threadIdx.x < world) {
        unsigned long long *flag = reinterpret_cast<unsigned long long *>(peers.base[threadIdx.x] + p2p_flags_offset(cap, world)) +
                                   ((size_t)phase * world + rank) * kP2pChunks + w;
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if ((int)threadIdx.x < world) {
        const unsigned long long *flag = reinterpret_cast<const unsigned long long *>(peers.base[rank] + p2p_flags_offset(cap, world)) +
                                         ((size_t)phase * world + threadIdx.x) * kP2pChunks + w;
        const long long t0 = (long long)wall_clock64();
        int spin = 0;
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            if ((++spin & 1023) == 0 && (long long)wall_clock64() - t0 > timeout_ticks) {   // the peer is gone (or stalled past the budget)
                __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                timed_out = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (spin) atomicMax(&wait_max, (unsigned long long)((long long)wall_clock64() - t0));
    }
    __syncthreads();
    __threadfence_system();
    if (wait_stats && threadIdx.x == 0) {   // peer-wait telemetry (pfa_p2p_wait_stats): this chunk's longest wait for a peer's flag
        atomicAdd(wait_stats + 2, wait_max);
        atomicAdd(wait_stats + 3, 1ull);
    }
    if (timed_out) {   // never hand back a partial sum
        T nan_v = (T)__builtin_nanf("");
        for (long long i = lo + threadIdx.x; i < hi; i += kP2pThreads) buf[i] = nan_v;
        return;
    }
    const char *mine = peers.base[rank];
    for (long long i = lo + threadIdx.x; i < hi; i += kP2pThreads) {
        T s = reinterpret_cast<const T *>(mine + p2p_slot_offset(cap, world, phase, 0))[i];
        for (int r = 1; r < world; ++r) s += __builtin_nontemporal_load(reinterpret_cast<const T *>(mine + p2p_slot_offset(cap, world, phase, r)) + i);
        buf[i] = s;
    }
}

// The flag-in-data exchange as a launch of its own (self-test at start-up, unit tests; the update runs it inside
// ppo_reduce_adam_kernel): one entry per float.
__global__ void __launch_bounds__(256) p2p_ll_all_reduce_kernel(LlArgs d, float *buf, unsigned n) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0) ll_status_exchange(d);   // as in the fused optimizer step: every exchange carries the ranks' status words
    if (i >= n) return;
    const float v = buf[i];
    ll_push(d, i, v);
    buf[i] = ll_wait_sum(d, i, v);
}

static bool g_p2p_enabled = true;   // pfa_p2p_enable: routing switch for A/B runs (the buffers stay mapped)
bool p2p_ll_ready(size_t entries_needed) { return g_p2p_enabled && g_p2p.ready && g_p2p.world > 1 && entries_needed <= g_p2p.ll_entries; }
LlArgs p2p_ll_next() {
    LlArgs d{};
    for (int q = 0; q < g_p2p.world; ++q) d.base[q] = reinterpret_cast<unsigned long long *>(g_p2p.peers.base[q] + g_p2p.ll_offset);
    d.rank = g_p2p.rank;
    d.world = g_p2p.world;
    d.entries = g_p2p.ll_entries;
    if (++g_p2p.ll_seq == 0) g_p2p.ll_seq = 2;   // 0 is the zeroed area's value; keep the phase parity going (2^32 calls away)
    d.seq = g_p2p.ll_seq;
    d.status = g_p2p.status;
    d.timeout_ticks = g_p2p.timeout_ticks;
    d.wait_stats = g_p2p.wait_stats;
    ++g_p2p.ll_calls;
    return d;
}
unsigned long long p2p_ll_calls() { return g_p2p.ll_calls; }

bool p2p_ready() { return g_p2p.ready; }
unsigned long long p2p_calls() { return g_p2p.calls; }
size_t p2p_capacity() { return g_p2p.ready ? g_p2p.cap_bytes : 0; }
int p2p_world() { return g_p2p.world; }
bool p2p_fits(size_t bytes) { return g_p2p_enabled && g_p2p.ready && bytes <= g_p2p.cap_bytes; }

int p2p_all_reduce(void *buf, size_t count, bool f64, hipStream_t stream) {
    PFA_REQUIRE(g_p2p.ready, "p2p: not initialised");
    const size_t bytes = count * (f64 ? 8 : 4);
    PFA_REQUIRE(bytes <= g_p2p.cap_bytes, "p2p: %zu bytes exceed the slot capacity %zu", bytes, g_p2p.cap_bytes);
    if (count == 0) return 0;
    const unsigned long long seq = ++g_p2p.seq;
    ++g_p2p.calls;
    const int phase = (int)(seq & 1);
    // Workgroups per call: a chunk is pushed to every peer by ONE workgroup, so the chunk size sets how many store streams run
    // next to each other on the links.  4 KB (10 workgroups for the MLP policy's 38 KB bucket);
    // at most kP2pChunks workgroups (a flag per source and chunk).
    constexpr long long chunk_bytes = 4096;
    long long chunks = (long long)((bytes + chunk_bytes - 1) / chunk_bytes);
    chunks = chunks < 1 ? 1 : (chunks > kP2pChunks ? kP2pChunks : chunks);
    ScopedKernelTimer timer("p2p_all_reduce", stream);
    if (f64)
        hipLaunchKernelGGL(p2p_all_reduce_kernel<double>, dim3((unsigned)chunks), dim3(kP2pThreads), 0, stream, g_p2p.peers, g_p2p.rank,
                           g_p2p.world, g_p2p.cap_bytes, phase, seq, (double *)buf, (long long)count, g_p2p.status, g_p2p.timeout_ticks, g_p2p.wait_stats);
    else
        hipLaunchKernelGGL(p2p_all_reduce_kernel<float>, dim3((unsigned)chunks), dim3(kP2pThreads), 0, stream, g_p2p.peers, g_p2p.rank,
                           g_p2p.world, g_p2p.cap_bytes, phase, seq, (float *)buf, (long long)count, g_p2p.status, g_p2p.timeout_ticks, g_p2p.wait_stats);
    PFA_LAUNCH_CHECK();
    return 0;
}

}  // namespace pfa

using namespace pfa;

// Allocate this rank's buffer (slots of `cap_bytes` per phase and source) and export its IPC handle (64 bytes).
extern "C" int pfa_p2p_alloc(int64_t cap_bytes, int32_t world, uint8_t *handle64_host) {
    PFA_REQUIRE(cap_bytes >= 256 && world >= 1 && world <= kP2pMaxRanks && handle64_host, "p2p.alloc: bad arguments (world must be 1..%d)", kP2pMaxRanks);
    PFA_REQUIRE(!g_p2p.ready && !g_p2p.local, "p2p.alloc: already allocated (pfa_p2p_close first)");
    const size_t cap = align_up((size_t)cap_bytes, 256);
    const size_t total = p2p_total_bytes(cap, world);
    void *p = nullptr;
    PFA_CHECK_HIP(hipExtMallocWithFlags(&p, total, hipDeviceMallocFinegrained));
    PFA_CHECK_HIP(hipMemset(p, 0, total));
    PFA_CHECK_HIP(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    PFA_CHECK_HIP(hipIpcGetMemHandle(&h, p));
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    std::memcpy(handle64_host, &h, 64);
    g_p2p.local = (char *)p;
    g_p2p.cap_bytes = cap;
    g_p2p.world = world;
    g_p2p.ll_offset = p2p_ll_offset(cap, world);
    g_p2p.ll_entries = p2p_ll_entries(cap);
    void *st = nullptr;
    PFA_CHECK_HIP(hipHostMalloc(&st, 64, hipHostMallocMapped));
    *(volatile int *)st = 0;
    g_p2p.status = (int *)st;
    long long ms = 30000;
    if (const char *e = std::getenv("PFA_WAIT_TIMEOUT_MS")) {
        const long long v = std::atoll(e);
        if (v > 0) ms = v;
    }
    g_p2p.timeout_ticks = ms * 100000;   // wall_clock64: 100 MHz
    void *wsp = nullptr;
    PFA_CHECK_HIP(hipMalloc(&wsp, 4 * sizeof(unsigned long long)));
    PFA_CHECK_HIP(hipMemset(wsp, 0, 4 * sizeof(unsigned long long)));
    g_p2p.wait_stats = (unsigned long long *)wsp;
    return 0;
}

// Map every peer's buffer.  `handles` = world x 64 bytes in rank order (this rank's own entry is ignored).
extern "C" int pfa_p2p_open(const uint8_t *handles_host, int32_t rank, int32_t world) {
    PFA_REQUIRE(handles_host && g_p2p.local && world == g_p2p.world && rank >= 0 && rank < world, "p2p.open: bad arguments / not allocated");
    for (int q = 0; q < world; ++q) {
        if (q == rank) {
            g_p2p.peers.base[q] = g_p2p.local;
            continue;
        }
        hipIpcMemHandle_t h;
        std::memcpy(&h, handles_host + (size_t)q * 64, 64);
        void *p = nullptr;
        PFA_CHECK_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        g_p2p.peers.base[q] = (char *)p;
    }
    g_p2p.rank = rank;
    g_p2p.seq = 0;
    g_p2p.ll_seq = 0;
    g_p2p.ready = true;
    return 0;
}

extern "C" int pfa_p2p_close(void) {
    if (g_p2p.ready)
        for (int q = 0; q < g_p2p.world; ++q)
            if (q != g_p2p.rank && g_p2p.peers.base[q]) (void)hipIpcCloseMemHandle(g_p2p.peers.base[q]);
    if (g_p2p.local) (void)hipFree(g_p2p.local);
    if (g_p2p.status) (void)hipHostFree(g_p2p.status);
    if (g_p2p.wait_stats) (void)hipFree(g_p2p.wait_stats);
    g_p2p = P2pState{};
    return 0;
}

extern "C" int pfa_p2p_status(void) {   // 0 ok, 1 a wait of THIS rank ran out, 2 a peer reported one (p2p_ll.hpp), -1 not initialised.  A plain host read.
    if (!g_p2p.ready || !g_p2p.status) return -1;
    return *(volatile int *)g_p2p.status;
}

// Recovery after a raised status word (a timed-out exchange): every rank calls pfa_p2p_seq(), the caller agrees on
// base >= every rank's value (e.g. MAX all-reduce over torch.distributed, + a margin), and every rank calls pfa_p2p_reset(base)
// between two barriers: the status word is cleared and both sequence counters restart from `base`, so nothing a late peer may still
// write with an old sequence number can satisfy a wait of the new epoch (flags and flag-in-data entries compare sequence numbers).
// The parameters of the replicas are NOT repaired here (a timed-out rank holds NaN): reload them (clean_pufferl.try_load_checkpoint)
// or re-broadcast them before training on.
extern "C" int64_t pfa_p2p_seq(void) {
    const unsigned long long a = g_p2p.seq, b = g_p2p.ll_seq;
    return (int64_t)(a > b ? a : b);
}
extern "C" int pfa_p2p_reset(int64_t base) {
    PFA_REQUIRE(g_p2p.ready && g_p2p.status, "p2p.reset: not initialised");
    PFA_REQUIRE(base >= pfa_p2p_seq() && base < 0xffffff00ll, "p2p.reset: base %lld must be >= this rank's sequence number %lld (and < 2^32)",
                (long long)base, (long long)pfa_p2p_seq());
    PFA_CHECK_HIP(hipDeviceSynchronize());
    *(volatile int *)g_p2p.status = 0;
    g_p2p.seq = (unsigned long long)base;
    g_p2p.ll_seq = (unsigned)base;
    return 0;
}
extern "C" int pfa_p2p_debug_set_status(int value) {   // tests: what a timed-out wait leaves behind, without waiting for one
    PFA_REQUIRE(g_p2p.ready && g_p2p.status, "p2p: not initialised");
    *(volatile int *)g_p2p.status = value;
    return 0;
}

extern "C" int pfa_p2p_all_reduce_f32(float *buf, int64_t count, pfa_stream_t stream) {
    PFA_REQUIRE(buf && count >= 0, "p2p.all_reduce: bad arguments");
    return p2p_all_reduce(buf, (size_t)count, false, (hipStream_t)stream);
}
extern "C" int pfa_p2p_all_reduce_f64(double *buf, int64_t count, pfa_stream_t stream) {
    PFA_REQUIRE(buf && count >= 0, "p2p.all_reduce: bad arguments");
    return p2p_all_reduce(buf, (size_t)count, true, (hipStream_t)stream);
}

// The flag-in-data form (csrc/p2p_ll.hpp) as a stand-alone all-reduce of up to `slot capacity / 4 + 2304` floats: what the fused
// optimizer-step kernel does per gradient entry, exposed for the start-up self-test and the unit tests.
extern "C" int pfa_p2p_ll_all_reduce_f32(float *buf, int64_t count, pfa_stream_t stream) {
    PFA_REQUIRE(buf && count >= 0, "p2p.ll_all_reduce: bad arguments");
    PFA_REQUIRE(g_p2p.ready, "p2p: not initialised");
    PFA_REQUIRE((size_t)count + 1 <= g_p2p.ll_entries, "p2p.ll_all_reduce: %lld floats exceed the %u entries of a slot (the last one carries the status)",
                (long long)count, g_p2p.ll_entries - 1);
    if (count == 0 || g_p2p.world == 1) return 0;
    const LlArgs d = p2p_ll_next();
    ScopedKernelTimer timer("p2p_ll_all_reduce", (hipStream_t)stream);
    hipLaunchKernelGGL(p2p_ll_all_reduce_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d, buf, (unsigned)count);
    PFA_LAUNCH_CHECK();
    return 0;
}
extern "C" int64_t pfa_p2p_ll_calls(void) { return (int64_t)g_p2p.ll_calls; }
// Peer-wait telemetry: out4 = { ticks, workgroups } of the flag-in-data exchanges (the optimizer steps inside ppo_reduce_adam_kernel),
// { ticks, chunks } of the flag-based all-reduces (the small f64 exchanges) — 100 MHz ticks a workgroup stood waiting for its slowest
// peer, summed since the last reset.  Synchronises the device (a 32-byte copy); reset != 0 clears the counters afterwards.
extern "C" int pfa_p2p_wait_stats(int64_t *out4_host, int reset) {
    PFA_REQUIRE(out4_host != nullptr, "p2p.wait_stats: null buffer");
    for (int i = 0; i < 4; ++i) out4_host[i] = 0;
    if (!g_p2p.wait_stats) return 0;
    PFA_CHECK_HIP(hipDeviceSynchronize());
    PFA_CHECK_HIP(hipMemcpy(out4_host, g_p2p.wait_stats, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) PFA_CHECK_HIP(hipMemset(g_p2p.wait_stats, 0, 4 * sizeof(unsigned long long)));
    return 0;
}
// Routing switch (bench.py's per-transport A/B, all ranks alike): 0 = pfa_dist_all_reduce_* and the native train loop stop using
// the peer path (they fall to the RCCL communicator) although it stays open; 1 = use it again.  Returns the previous setting.
extern "C" int pfa_p2p_enable(int on) {
    const int was = g_p2p_enabled ? 1 : 0;
    g_p2p_enabled = on != 0;
    return was;
}
