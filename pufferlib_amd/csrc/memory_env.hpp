// memory_env.hpp — state layout and per-env step of the device-resident ocean `Memory` vecenv (ocean.py:65-123), shared by the
// protocol kernels (memory.hip) and the fused recurrent rollout (lstm_fused.hip).
#pragma once
#include "common.hpp"
#include "episode_fin.hpp"
#include "mt19937.hpp"

namespace pfa {

constexpr int kMemDP = 16;        // observation row stride in floats (1 real column)
constexpr int kMemMaxLen = 16;    // mem_length limit (digits kept as bits)
constexpr int kMemTapeThreads = 640;

struct MemoryHeader {
    uint32_t mt[kMtN];   // raw state of the CURRENT block of the shared stream
    int mt_idx;          // next word of the current block (>= 624: regenerate first)
    int skip;            // words of the shared stream already consumed by env N-1's own first reset
    int underrun;        // set if a reset found no tape round (host bookkeeping error)
    int pad;
    long long rounds_filled;
};
struct MemoryEnv {
    int tick, done, ep_length;
    uint32_t sol_bits;   // bit j = solution[j], j < L
    int all_correct;     // every digit submitted so far matched
    int pad;
    long long rounds;    // reset rounds this env has consumed
    double ep_return;
};
struct MemoryView {
    MemoryHeader *hdr;
    MemoryEnv *env;
    EpisodeFin *fin;
    uint32_t *tape;      // [tape_rounds][n]
    uint32_t *first;     // [n] solution bits of the episode async_reset starts (drawn from the env's own seed)
    int n, L, D, H, tape_rounds;
};
__host__ __device__ inline size_t memory_state_bytes(int n, int tape_rounds) {
    return sizeof(MemoryHeader) + (size_t)n * (sizeof(MemoryEnv) + sizeof(EpisodeFin)) + ((size_t)tape_rounds + 1) * n * sizeof(uint32_t);
}
__host__ __device__ inline MemoryView memory_view(void *state, const pfa_memory_config &c) {
    MemoryView v;
    char *p = (char *)state;
    v.hdr = (MemoryHeader *)p;
    p += sizeof(MemoryHeader);
    v.env = (MemoryEnv *)p;
    p += (size_t)c.num_envs * sizeof(MemoryEnv);
    v.fin = (EpisodeFin *)p;
    p += (size_t)c.num_envs * sizeof(EpisodeFin);
    v.tape = (uint32_t *)p;
    p += (size_t)c.tape_rounds * c.num_envs * sizeof(uint32_t);
    v.first = (uint32_t *)p;
    v.n = c.num_envs;
    v.L = c.mem_length;
    v.D = c.mem_delay;
    v.H = 2 * c.mem_length + c.mem_delay;
    v.tape_rounds = c.tape_rounds;
    return v;
}

__device__ __forceinline__ void memory_begin_episode(MemoryEnv &s, uint32_t bits, float &obs, float &reward, bool &terminal) {
    s.tick = 1;
    s.done = 0;
    s.ep_length = 0;
    s.ep_return = 0.0;
    s.sol_bits = bits;
    s.all_correct = 1;
    obs = (float)(bits & 1u);  // solution[0]
    reward = 0.0f;
    terminal = false;
}

// ocean.py:102-123 + postprocess.py:22-54 + emulation.py:194-228 for one env; returns true when the episode finished
__device__ __forceinline__ bool memory_step(const MemoryView &v, MemoryEnv &s, int action, float &obs, float &reward, bool &terminal,
                                            double &fin_return, int &fin_length, double &fin_score) {
    float ob = 0.0f;
    double r = 0.0;
    if (s.tick < v.L) {
        ob = (float)((s.sol_bits >> s.tick) & 1u);
        r = action == 0 ? 1.0 : 0.0;
    }
    if (s.tick >= v.L + v.D) {
        const int idx = s.tick - v.L - v.D;
        const int ok = action == (int)((s.sol_bits >> idx) & 1u);
        r = ok ? 1.0 : 0.0;
        s.all_correct &= ok;
    }
    s.tick += 1;
    terminal = s.tick == v.H;
    s.ep_return += r;
    s.ep_length += 1;
    s.done = terminal;
    obs = ob;
    reward = (float)r;
    if (terminal) {
        fin_return = s.ep_return;
        fin_length = s.ep_length;
        fin_score = s.all_correct ? 1.0 : 0.0;
    }
    return terminal;
}

}  // namespace pfa
