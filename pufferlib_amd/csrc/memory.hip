// memory.hip — ocean `Memory` (pufferlib/environments/ocean/ocean.py:65-123) as a device-resident vecenv: the third env
// family and the one that needs a recurrent policy.  Reference stack per env: pufferlib.vector.Serial
// (vector.py:78-162) over make_memory (ocean/environment.py:41-44) = GymnasiumPufferEnv + EpisodeStats + ocean.Memory.
//
// Episode (horizon H = 2L + D): reset shows solution[0]; for ticks < L the observation is the next digit and reward =
// [action == 0]; D delay ticks with nothing; for ticks >= L + D reward = [action == solution[tick - L - D]]; terminal when
// tick == H (H - 1 steps), score = [all L digits repeated correctly]; the next send() is the auto-reset row.
//
// Randomness: every reset draws solution = np.random.randint(0, 2, size=H) from numpy's PROCESS-GLOBAL legacy generator —
// MT19937 seeded by init_genrand, one 32-bit word per element (range 1: mask 1, never rejected), bit 0 used.  async_reset
// seeds the generator per env (seed + i) right before that env's draw, so initial solutions depend on the env's seed only
// and the stream every later reset shares is the one env N-1 left behind, H words in.  All envs finish on the same send,
// so reset round r hands env e the words [r*N*H + e*H, +H) of that stream.  As for Squared (squared.hip) the stream does
// not depend on actions and is drawn ahead of time into a tape (one uint32 of solution bits per (round, env)).
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

// np.random.seed(seed + e) then the first reset's draw, for every env (one thread each): only block-1 words j < L matter
// (solution[L:] is overwritten with -1, ocean.py:96), word j = temper(twist(init[j], init[j+1], init[j+397])), so the thread
// walks init_genrand's recurrence once and keeps the few words it needs.  The thread of env N-1 also leaves the complete
// initial state in the header: that is the stream all later resets share.  Two kernels: the stream arithmetic, then the
// plain per-env initialisation.
__global__ void __launch_bounds__(256) memory_seed_kernel(MemoryView v, long long seed) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= v.n) return;
    const bool last = e == v.n - 1;
    uint32_t x = (uint32_t)(seed + e);
    uint32_t prev = x;       // init[i-1]
    uint32_t bits = 0;
    // word j of block 1 needs init[j], init[j+1], init[j+397]; walk once to index 397+L-1 keeping y_j = (init[j]&U)|(init[j+1]&L)
    uint32_t y[kMemMaxLen];
#pragma unroll
    for (int j = 0; j < kMemMaxLen; ++j) y[j] = 0;
    if (last) v.hdr->mt[0] = x;
    const int stop = last ? kMtN : kMtM + v.L;
    for (int i = 1; i < stop; ++i) {
        x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
        if (last) v.hdr->mt[i] = x;
#pragma unroll
        for (int j = 0; j < kMemMaxLen; ++j) {
            if (i == j + 1) y[j] = (prev & 0x80000000u) | (x & 0x7fffffffu);
            if (i == j + kMtM && j < v.L) {
                const uint32_t w = x ^ (y[j] >> 1) ^ ((y[j] & 1u) ? 0x9908b0dfu : 0u);
                bits |= (mt_temper(w) & 1u) << j;
            }
        }
        prev = x;
    }
    v.first[e] = bits;
    if (last) {
        v.hdr->mt_idx = kMtN;  // numpy regenerates on the first draw
        v.hdr->skip = v.H;     // ... and env N-1 took the first H words itself
        v.hdr->underrun = 0;
        v.hdr->rounds_filled = 0;
    }
}

__global__ void __launch_bounds__(256) memory_begin_kernel(MemoryView v, float *obs, float *rewards, uint8_t *terminals,
                                                          uint8_t *truncations, uint8_t *masks) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= v.n) return;
    MemoryEnv s;
    float o, r;
    bool t;
    memory_begin_episode(s, v.first[e], o, r, t);
    s.pad = 0;
    s.rounds = 0;
    v.env[e] = s;
    EpisodeFin f = {};
    v.fin[e] = f;
#pragma unroll
    for (int k = 0; k < kMemDP; ++k) obs[(size_t)e * kMemDP + k] = 0.0f;
    obs[(size_t)e * kMemDP] = o;
    rewards[e] = 0.0f;
    terminals[e] = 0;
    truncations[e] = 0;
    masks[e] = 1;
}

// Draw `rounds` more reset rounds (rounds * N * H words of the shared stream) into the tape ring.  One workgroup; the MT
// state is double-buffered in LDS and regenerated by all threads (mt_next_block); every word of a block is placed in
// parallel: word at stream position pos -> (round, env, digit) = (pos / (N*H), pos % (N*H) / H, pos % H).
__global__ void __launch_bounds__(kMemTapeThreads) memory_tape_kernel(MemoryView v, int rounds) {
    __shared__ uint32_t mt[2][kMtN];
    __shared__ uint32_t out[kMtN];
    const int tid = threadIdx.x;
    const long long first_round = v.hdr->rounds_filled;
    for (long long i = tid; i < (long long)rounds * v.n; i += kMemTapeThreads) {
        const long long round = first_round + i / v.n;
        v.tape[(size_t)(round % v.tape_rounds) * v.n + (size_t)(i % v.n)] = 0u;
    }
    if (tid < kMtN) {
        mt[0][tid] = v.hdr->mt[tid];
        out[tid] = mt_temper(mt[0][tid]);
    }
    int cur = 0, idx = v.hdr->mt_idx;
    long long skip = v.hdr->skip;
    const long long per_round = (long long)v.n * v.H;
    const long long need = (long long)rounds * per_round;
    long long produced = 0;
    __syncthreads();
    __threadfence_block();
    while (produced < need) {
        if (idx >= kMtN) {
            mt_next_block(mt[cur], mt[cur ^ 1], out);
            cur ^= 1;
            idx = 0;
        }
        if (skip > 0) {  // words env N-1 consumed at async_reset
            const long long take = skip < kMtN - idx ? skip : kMtN - idx;
            idx += (int)take;
            skip -= take;
            continue;
        }
        const long long avail = kMtN - idx, left = need - produced;
        const int take = (int)(avail < left ? avail : left);
        if (tid < take) {
            const long long pos = produced + tid;
            const long long round = first_round + pos / per_round;
            const long long rem = pos % per_round;
            const int env = (int)(rem / v.H), digit = (int)(rem % v.H);
            if (digit < v.L && (out[idx + tid] & 1u))
                atomicOr(&v.tape[(size_t)(round % v.tape_rounds) * v.n + env], 1u << digit);
        }
        idx += take;
        produced += take;
        __syncthreads();
    }
    __syncthreads();
    if (tid < kMtN) v.hdr->mt[tid] = mt[cur][tid];
    if (tid == 0) {
        v.hdr->mt_idx = idx;
        v.hdr->skip = (int)skip;
        v.hdr->rounds_filled = first_round + rounds;
    }
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

__global__ void __launch_bounds__(256) memory_send_kernel(MemoryView v, const long long *actions, float *obs, float *rewards,
                                                         uint8_t *terminals, uint8_t *truncations, uint8_t *masks) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= v.n) return;
    MemoryEnv s = v.env[e];
    float o, r;
    bool t;
    v.fin[e].last_fin = 0;
    if (s.done) {
        if (s.rounds >= v.hdr->rounds_filled) v.hdr->underrun = 1;
        const uint32_t bits = v.tape[(size_t)(s.rounds % v.tape_rounds) * v.n + e];
        const long long rounds = s.rounds + 1;
        memory_begin_episode(s, bits, o, r, t);
        s.rounds = rounds;
    } else {
        double fr, fs;
        int fl;
        if (memory_step(v, s, (int)actions[e], o, r, t, fr, fl, fs)) episode_account(v.fin[e], fr, fl, fs);
    }
    v.env[e] = s;
    obs[(size_t)e * kMemDP] = o;
    rewards[e] = r;
    terminals[e] = t ? 1 : 0;
    truncations[e] = 0;
    masks[e] = 1;
}

__global__ void memory_debug_solution_kernel(MemoryView v, uint32_t *bits_out, int *underrun_out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < v.n) bits_out[e] = v.env[e].sol_bits;
    if (e == 0) *underrun_out = v.hdr->underrun;
}

static int check_memory_config(const pfa_memory_config *c) {
    PFA_REQUIRE(c != nullptr, "memory: null config");
    PFA_REQUIRE(c->num_envs >= 1, "memory: num_envs must be >= 1");
    PFA_REQUIRE(c->mem_length >= 1 && c->mem_length <= kMemMaxLen, "memory: mem_length must be in 1..%d (got %d)", kMemMaxLen, c->mem_length);
    PFA_REQUIRE(c->mem_delay >= 0 && 2 * c->mem_length + c->mem_delay <= 1024, "memory: mem_delay out of range");
    PFA_REQUIRE(c->tape_rounds >= 2, "memory: tape_rounds must be >= 2");
    return 0;
}

}  // namespace pfa

using namespace pfa;

extern "C" size_t pfa_memory_state_bytes(const pfa_memory_config *cfg) {
    if (check_memory_config(cfg)) return 0;
    return memory_state_bytes(cfg->num_envs, cfg->tape_rounds);
}

extern "C" int pfa_memory_async_reset(void *state, const pfa_memory_config *cfg, int64_t seed, float *obs, float *rewards,
                                      uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    if (int rc = check_memory_config(cfg)) return rc;
    PFA_REQUIRE(state && obs && rewards && terminals && truncations && masks, "memory.async_reset: null buffer");
    PFA_REQUIRE(seed >= 0 && seed + cfg->num_envs - 1 <= 0xFFFFFFFFll, "memory.async_reset: np.random.seed needs 0 <= seed < 2**32");
    hipLaunchKernelGGL(memory_seed_kernel, dim3((unsigned)((cfg->num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       memory_view(state, *cfg), (long long)seed);
    PFA_LAUNCH_CHECK();
    hipLaunchKernelGGL(memory_begin_kernel, dim3((unsigned)((cfg->num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       memory_view(state, *cfg), obs, rewards, terminals, truncations, masks);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_memory_fill_tape(void *state, const pfa_memory_config *cfg, int32_t rounds, pfa_stream_t stream) {
    if (int rc = check_memory_config(cfg)) return rc;
    PFA_REQUIRE(state && rounds >= 0 && rounds <= cfg->tape_rounds, "memory.fill_tape: rounds out of range");
    if (rounds == 0) return 0;
    ScopedKernelTimer timer("memory_tape", (hipStream_t)stream);
    hipLaunchKernelGGL(memory_tape_kernel, dim3(1), dim3(kMemTapeThreads), 0, (hipStream_t)stream, memory_view(state, *cfg), (int)rounds);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_memory_send(void *state, const pfa_memory_config *cfg, const int64_t *actions, float *obs, float *rewards,
                               uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    if (int rc = check_memory_config(cfg)) return rc;
    PFA_REQUIRE(state && actions && obs && rewards && terminals && truncations && masks, "memory.send: null buffer");
    hipLaunchKernelGGL(memory_send_kernel, dim3((unsigned)((cfg->num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       memory_view(state, *cfg), (const long long *)actions, obs, rewards, terminals, truncations, masks);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_memory_episode_stats(void *state, const pfa_memory_config *cfg, double *out4, int32_t reset, pfa_stream_t stream) {
    if (int rc = check_memory_config(cfg)) return rc;
    PFA_REQUIRE(state && out4, "memory.episode_stats: null buffer");
    hipLaunchKernelGGL(episode_stats_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, memory_view(state, *cfg).fin, (int)cfg->num_envs,
                       out4, (int)reset, (const int *)&memory_view(state, *cfg).hdr->underrun);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_memory_last_infos(void *state, const pfa_memory_config *cfg, uint8_t *finished, double *episode_return,
                                     int32_t *episode_length, double *score, pfa_stream_t stream) {
    if (int rc = check_memory_config(cfg)) return rc;
    PFA_REQUIRE(state && finished && episode_return && episode_length && score, "memory.last_infos: null buffer");
    hipLaunchKernelGGL(episode_infos_kernel, dim3((unsigned)((cfg->num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       memory_view(state, *cfg).fin, (int)cfg->num_envs, finished, episode_return, (int *)episode_length, score);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_memory_debug_solutions(void *state, const pfa_memory_config *cfg, uint32_t *bits, int32_t *underrun,
                                          pfa_stream_t stream) {
    if (int rc = check_memory_config(cfg)) return rc;
    PFA_REQUIRE(state && bits && underrun, "memory.debug_solutions: null buffer");
    hipLaunchKernelGGL(memory_debug_solution_kernel, dim3((unsigned)((cfg->num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       memory_view(state, *cfg), bits, (int *)underrun);
    PFA_LAUNCH_CHECK();
    return 0;
}
