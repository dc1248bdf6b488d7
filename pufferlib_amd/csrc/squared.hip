// squared.hip — vecenv kernels for ocean Squared behind the pufferlib.vector backend protocol
// (async_reset / send; pufferlib/vector.py:112-156) and the reset-target tape that reproduces the
// process-global `random.sample` stream (ocean.py:449-459) bit-exactly.
//
// Roofline: integer state machines, one thread per env, ~300 B touched per env per step -> HBM/latency
// bound; nothing here is GEMM-shaped.
#include "common.hpp"
#include "mt19937.hpp"
#include "squared_env.hpp"

namespace pfa {

// ---------------------------------------------------------------------------------------------
// random.sample(range(n), k) as a word-at-a-time state machine (Lib/random.py Random.sample +
// _randbelow_with_getrandbits).  feed() consumes ONE tempered 32-bit output and returns true when the
// sample is complete.  The pool-swap branch (n <= setsize) is kept as an override list
// (position -> value) instead of an n-entry pool.
// ---------------------------------------------------------------------------------------------
struct Sampler {
    int n, k, i;
    bool pool_path;
    uint8_t picks[kMaxTargets];   // chosen population indices
    uint8_t ov_pos[kMaxTargets];  // pool overrides
    uint8_t ov_val[kMaxTargets];

    __device__ void begin(int n_, int k_) {
        n = n_;
        k = k_;
        i = 0;
        int setsize = 21;
        if (k > 5) {
            int p = 1;
            while (p < 3 * k) p *= 4;
            setsize += p;
        }
        pool_path = n <= setsize;
    }
    __device__ int lookup(int pos, int upto) const {
        for (int q = upto - 1; q >= 0; --q)
            if (ov_pos[q] == pos) return ov_val[q];
        return pos;
    }
    __device__ bool feed(uint32_t word) {
        const int m = pool_path ? n - i : n;
        const int bits = 32 - __clz(m);  // m.bit_length(), m >= 1
        const int r = (int)(word >> (32 - bits));
        if (r >= m) return false;
        if (pool_path) {
            picks[i] = (uint8_t)lookup(r, i);
            ov_val[i] = (uint8_t)lookup(m - 1, i);
            ov_pos[i] = (uint8_t)r;
        } else {
            for (int q = 0; q < i; ++q)
                if (picks[q] == r) return false;
            picks[i] = (uint8_t)r;
        }
        ++i;
        return i == k;
    }
};

// population index -> cell (x*g+y) of possible_targets (ocean.py:444-446): row-major perimeter.
__device__ __forceinline__ int perimeter_cell(int idx, int g) {
    if (idx < g) return idx;                       // x = 0, y = idx
    const int last_row_start = g + 2 * (g - 2);    // entries before x = g-1
    if (idx >= last_row_start) return (g - 1) * g + (idx - last_row_start);
    const int q = idx - g;                         // rows 1..g-2 contribute (x,0),(x,g-1)
    const int x = 1 + (q >> 1);
    return x * g + ((q & 1) ? g - 1 : 0);
}

// ---------------------------------------------------------------------------------------------
// async_reset: env i runs random.seed(seed+i); reset()  (vector.py:129-130, ocean.py:448-463)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) squared_seed_kernel(SquaredView v, long long seed, float *obs, float *rewards,
                                                          uint8_t *terminals, uint8_t *truncations, uint8_t *masks) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= v.n) return;
    uint32_t *mt = v.seed_mt + e;
    const size_t stride = (size_t)v.n;
    long long s = seed + e;
    mt_seed_strided(mt, stride, (uint64_t)(s < 0 ? -s : s));
    int idx = kMtN;
    unsigned long long words = 0;
    Sampler sm;
    sm.begin(8 * v.d, v.nt);
    for (;;) {
        if (idx >= kMtN) {
            mt_regenerate_strided(mt, stride);
            idx = 0;
        }
        const uint32_t w = mt_temper(mt[(size_t)idx * stride]);
        ++idx;
        ++words;
        if (sm.feed(w)) break;
    }
    float *grid = obs + (size_t)e * v.stride;
    for (int i = 0; i < v.stride; ++i) grid[i] = 0.0f;
    grid[v.d * v.g + v.d] = -1.0f;
    for (int t = 0; t < v.nt; ++t) {
        const int c = perimeter_cell(sm.picks[t], v.g);
        v.tgt[(size_t)t * v.n + e] = (uint16_t)c;
        grid[c] = 1.0f;
    }
    v.ax[e] = (int8_t)v.d;
    v.ay[e] = (int8_t)v.d;
    v.tick[e] = 0;
    v.done[e] = 0;
    v.rem[e] = v.nt >= 32 ? 0xffffffffu : ((1u << v.nt) - 1u);
    v.rounds[e] = 0;
    v.ep_ret[e] = 0.0;
    v.ep_len[e] = 0;
    v.fin[e] = 0;
    v.acc_cnt[e] = 0;
    v.acc_ret[e] = 0.0;
    v.acc_len[e] = 0;
    v.acc_score[e] = 0.0;
    rewards[e] = 0.0f;
    terminals[e] = 0;
    truncations[e] = 0;
    masks[e] = 1;
    if (e == v.n - 1) {  // the generator every later unseeded reset() draws from (SURVEY.md hard part 1)
        for (int i = 0; i < kMtN; ++i) v.hdr->mt[i] = mt[(size_t)i * stride];
        v.hdr->mt_idx = idx;
        v.hdr->underrun = 0;
        v.hdr->words = words;
        v.hdr->rounds_filled = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// Tape fill: draw `rounds` x N samples from the shared stream, in env order.
// One workgroup of 640 threads (one per word of the padded 624-word MT block).  Per block:
//   3 dependent phases of the MT recurrence into the OTHER state buffer (double-buffered, so a phase only needs
//   the barrier that publishes the previous phase) with the tempering fused into the store  -> 3 barriers
//   nt == 1: every word is tested in parallel; accepted draws are compacted with ballot + per-wave counts -> 2 barriers
//   general: thread 0 walks the block feeding the resumable Sampler.
// The stream does not depend on actions, so the trainer runs this on a side stream under the previous rollout.
// ---------------------------------------------------------------------------------------------
constexpr int kTapeThreads = 640;
constexpr int kTapeWaves = kTapeThreads / 64;

__global__ void __launch_bounds__(kTapeThreads) squared_tape_kernel(SquaredView v, int rounds) {
    __shared__ uint32_t mt[2][kMtN];
    __shared__ uint32_t out[kMtN];
    __shared__ int wave_cnt[kTapeWaves];
    __shared__ int s_idx_end;
    __shared__ Sampler s_sm;
    __shared__ long long s_produced;
    __shared__ int s_idx;
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
    int cur = 0;
    if (tid < kMtN) {
        mt[0][tid] = v.hdr->mt[tid];
        out[tid] = mt_temper(mt[0][tid]);
    }
    int idx = v.hdr->mt_idx;
    const long long first_round = v.hdr->rounds_filled;
    const long long need = (long long)rounds * v.n;  // samples to draw
    long long produced = 0;
    unsigned long long words = 0;
    const int n_pop = 8 * v.d;
    const int bits = 32 - __clz(n_pop);
    if (tid == 0) s_sm.begin(n_pop, v.nt);
    __syncthreads();
    while (produced < need) {
        if (idx >= kMtN) {
            mt_next_block(mt[cur], mt[cur ^ 1], out);
            cur ^= 1;
            idx = 0;
        }
        if (v.nt == 1) {
            // parallel: word w accepted iff (out[w] >> (32-bits)) < n_pop; sample index = running count
            if (tid == 0) s_idx_end = kMtN;
            int r = -1;
            if (tid >= idx && tid < kMtN) {
                const int cnd = (int)(out[tid] >> (32 - bits));
                if (cnd < n_pop) r = cnd;
            }
            const unsigned long long bal = __ballot(r >= 0);
            if (lane == 0) wave_cnt[wv] = __popcll(bal);
            __syncthreads();
            int before = 0, total = 0;
#pragma unroll
            for (int q = 0; q < kTapeWaves; ++q) {
                const int cq = wave_cnt[q];
                before += q < wv ? cq : 0;
                total += cq;
            }
            const long long pos = produced + before + __popcll(bal & ((1ull << lane) - 1ull));
            if (r >= 0 && pos < need) {
                const long long round = first_round + pos / v.n;
                const int env = (int)(pos % v.n);
                v.tape[((size_t)(round % v.tape_rounds) * v.nt) * v.n + env] = (uint16_t)perimeter_cell(r, v.g);
                if (pos == need - 1) s_idx_end = tid + 1;
            }
            __syncthreads();
            const int idx_end = s_idx_end;
            words += (unsigned long long)(idx_end - idx);
            idx = idx_end;
            produced = produced + total < need ? produced + total : need;
        } else {
            // general: thread 0 feeds words to the resumable sampler until the block or the job ends
            if (tid == 0) {
                long long p = produced;
                int i = idx;
                while (i < kMtN && p < need) {
                    if (s_sm.feed(out[i++])) {
                        const long long round = first_round + p / v.n;
                        const int env = (int)(p % v.n);
                        for (int t = 0; t < v.nt; ++t)
                            v.tape[((size_t)(round % v.tape_rounds) * v.nt + t) * v.n + env] =
                                (uint16_t)perimeter_cell(s_sm.picks[t], v.g);
                        ++p;
                        s_sm.begin(n_pop, v.nt);
                    }
                }
                s_produced = p;
                s_idx = i;
            }
            __syncthreads();
            words += (unsigned long long)(s_idx - idx);
            produced = s_produced;
            idx = s_idx;
            __syncthreads();
        }
    }
    __syncthreads();
    if (tid < kMtN) v.hdr->mt[tid] = mt[cur][tid];
    if (tid == 0) {
        v.hdr->mt_idx = idx;
        v.hdr->words += words;
        v.hdr->rounds_filled = first_round + rounds;
    }
}

// ---------------------------------------------------------------------------------------------
// Tape fill, single-target form (num_targets == 1: random.sample(pop, 1) is ONE _randbelow, so word w yields a draw iff
// (w >> (32 - bits)) < n_pop, independently of every other word).  The work then splits into
//   (1) the only sequential part: the raw MT19937 words.  x[k+624] = twist(x[k], x[k+1], x[k+397]) lets 227 consecutive
//       words be computed at once; ONE small workgroup slides that 227-word window through LDS, two steps per barrier (one extra
//       wavefront on the SIMDs of a single CU: it runs under the rollout without slowing it) and streams the raw words to scratch —
//       `squared_tape_words`;
//   (2) everything else in parallel over all words: temper, test, count (`squared_tape_count`), then exclusive offsets,
//       compaction into (round, env) tape slots, and the stream state after the last word consumed (`squared_tape_select`).
// The number of words the draws will need is only known statistically (acceptance n_pop / 2^bits), so (1) produces the
// expectation + 2 % + 6 blocks (>= 15 sigma at every size); if the words still run out, `underrun` is raised — never silent.
// Scratch = the per-env seeding area `seed_mt` (u32[624][N]), idle between async_resets.
// ---------------------------------------------------------------------------------------------
constexpr int kMtStep = kMtN - kMtM;        // 227
constexpr int kTapeLin = 12288;             // words of LDS the raw stream slides through (48 KB)
constexpr int kSelWords = 16;               // consecutive words per thread in the parallel passes
constexpr int kSelThreads = 256;
constexpr int kSelChunk = kSelWords * kSelThreads;

struct TapeJob {          // written by the words kernel, read by the parallel passes (hdr itself is rewritten by the last pass)
    long long first_round;
    long long need;       // draws to produce
    long long avail;      // words available from the stream position on
    int idx0;             // index of the next word inside block 0 (may be 624)
    int blocks;           // regenerated blocks after block 0
};

// x[n] = x[n - 227] ^ g(x[n - 624], x[n - 623]): the word 227 back enters with a plain XOR, everything else depends on words at
// least 397 back.  One SLOT per thread (227 slots = the words of one step; four wavefronts, the last with 35 lanes) and TWO steps
// per barrier: both steps' g operands lie in the 624 words in front of the super-step (454 + 1 of them), the second step's
// XOR operand is the first step's word of the same slot (a register), and so is the next super-step's.  The chain per 454 words
// is then one LDS round trip (store -> barrier -> load), g, two XORs — instead of a single wavefront issuing every load and store
// of two 227-word steps itself (round 5: 209 us for a 128-step rollout's words, hidden only while a train() follows every
// evaluate()).  The stream still slides LINEARLY through LDS (x[base + i] at lin[w + i]); when the
// window reaches the end its last 624 words move back to the front.
constexpr int kTapeWordThreads = 256;
__device__ __forceinline__ uint32_t mt_g(uint32_t cur, uint32_t nxt) {   // mt_twist(cur, nxt, far) == far ^ mt_g(cur, nxt)
    const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__global__ void __launch_bounds__(kTapeWordThreads) squared_tape_words_kernel(SquaredView v, uint32_t *raw, TapeJob *job, int rounds, int blocks) {
    __shared__ uint32_t lin[kTapeLin];
    const int tid = threadIdx.x;
    for (int i = tid; i < kMtN; i += kTapeWordThreads) {
        const uint32_t w = v.hdr->mt[i];
        lin[i] = w;
        raw[i] = w;
    }
    if (tid == 0) {
        const int idx0 = v.hdr->mt_idx;
        job->first_round = v.hdr->rounds_filled;
        job->need = (long long)rounds * v.n;
        job->idx0 = idx0;
        job->blocks = blocks;
        job->avail = (long long)(blocks + 1) * kMtN - idx0;
    }
    lds_barrier();
    // new words x[624 + k], k in [0, blocks * 624), in whole super-steps of 2 x 227 (the last one may run past the end: the scratch
    // has the slack, the extra words are simply more of the same stream)
    const int supers = (blocks * kMtN + 2 * kMtStep - 1) / (2 * kMtStep);
    const bool own = tid < kMtStep;
    uint32_t far = own ? lin[tid + kMtM] : 0u;     // x[base + tid + 397]: from here on the word this slot produced one step earlier
    uint32_t *out = raw + kMtN + tid;
    int w = 0;                                     // window start inside lin (uniform)
    for (int ss = 0; ss < supers; ++ss) {
        if (own) {
            const uint32_t *src = lin + w + tid;
            const uint32_t c0 = src[0], n0 = src[1], c1 = src[kMtStep], n1 = src[kMtStep + 1];
            const uint32_t a = far ^ mt_g(c0, n0);
            const uint32_t b = a ^ mt_g(c1, n1);
            lin[w + kMtN + tid] = a;
            lin[w + kMtN + kMtStep + tid] = b;
            out[0] = a;
            out[kMtStep] = b;
            far = b;
        }
        w += 2 * kMtStep;
        out += 2 * kMtStep;
        lds_barrier();
        if (w + kMtN + 2 * kMtStep + 8 > kTapeLin) {       // (uniform) move the window back to the front: [w, w + 624) and [0, 624) are disjoint
            for (int i = tid; i < kMtN; i += kTapeWordThreads) lin[i] = lin[w + i];
            w = 0;
            lds_barrier();
        }
    }
}

__device__ __forceinline__ uint32_t tape_accept_mask(const uint32_t *raw, const TapeJob &job, long long p0, int bits, int n_pop,
                                                     uint32_t cand[kSelWords]) {
    // words p0 .. p0+15 of the stream (position p <-> raw[idx0 + p]); bit i set <=> word p0+i exists and is accepted
    uint32_t mask = 0;
#pragma unroll
    for (int i = 0; i < kSelWords; ++i) {
        const long long p = p0 + i;
        uint32_t c = 0xffffffffu;
        if (p < job.avail) c = mt_temper(raw[job.idx0 + p]) >> (32 - bits);
        cand[i] = c;
        if (c < (uint32_t)n_pop) mask |= 1u << i;
    }
    return mask;
}

__global__ void __launch_bounds__(kSelThreads) squared_tape_count_kernel(SquaredView v, const uint32_t *raw, const TapeJob *jobp,
                                                                         int *counts) {
    __shared__ int wsum[kSelThreads / 64];
    const TapeJob job = *jobp;
    const int n_pop = 8 * v.d, bits = 32 - __clz(n_pop);
    uint32_t cand[kSelWords];
    const long long p0 = (long long)blockIdx.x * kSelChunk + (long long)threadIdx.x * kSelWords;
    int c = __popc(tape_accept_mask(raw, job, p0, bits, n_pop, cand));
    for (int off = 32; off; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ void __launch_bounds__(kSelThreads) squared_tape_select_kernel(SquaredView v, const uint32_t *raw, const TapeJob *jobp,
                                                                          const int *counts, int rounds) {
    __shared__ long long s_red[kSelThreads / 64];
    __shared__ int s_wave[kSelThreads / 64];
    __shared__ long long s_last;     // stream position of the word that produced the last draw (-1: not in this workgroup)
    const TapeJob job = *jobp;
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
    const int n_pop = 8 * v.d, bits = 32 - __clz(n_pop);
    // draws produced by the workgroups before this one
    long long before = 0;
    for (int i = tid; i < (int)blockIdx.x; i += kSelThreads) before += counts[i];
    for (int off = 32; off; off >>= 1) before += __shfl_xor(before, off, 64);
    if (lane == 0) s_red[wv] = before;
    if (tid == 0) s_last = -1;
    __syncthreads();
    before = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    const bool tail = blockIdx.x == gridDim.x - 1;
    if (before >= job.need) return;   // (uniform) the stream position was fixed by an earlier workgroup
    uint32_t cand[kSelWords];
    const long long p0 = (long long)blockIdx.x * kSelChunk + (long long)tid * kSelWords;
    const uint32_t mask = tape_accept_mask(raw, job, p0, bits, n_pop, cand);
    // exclusive prefix of the per-thread counts over the workgroup
    const int mine = __popc(mask);
    int incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    int wave_base = 0, total = 0;
    for (int q = 0; q < kSelThreads / 64; ++q) {
        wave_base += q < wv ? s_wave[q] : 0;
        total += s_wave[q];
    }
    long long pos = before + wave_base + incl - mine;
#pragma unroll
    for (int i = 0; i < kSelWords; ++i) {
        if ((mask >> i) & 1u) {
            if (pos < job.need) {
                const long long round = job.first_round + pos / v.n;
                const int env = (int)(pos % v.n);
                v.tape[(size_t)(round % v.tape_rounds) * v.n + env] = (uint16_t)perimeter_cell((int)cand[i], v.g);
                if (pos == job.need - 1) s_last = p0 + i;
            }
            ++pos;
        }
    }
    __syncthreads();
    const long long last = s_last;
    if (last >= 0) {
        // the stream after the last consumed word: the block that holds the next word and the index inside it (an index of 624
        // = "regenerate first" stays on the old block, as CPython and the one-workgroup kernel leave it)
        const long long consumed = last + 1, at = job.idx0 + consumed;
        long long blk = at / kMtN;
        int idx = (int)(at % kMtN);
        if (idx == 0 && blk > 0) { blk -= 1; idx = kMtN; }
        for (int i = tid; i < kMtN; i += kSelThreads) v.hdr->mt[i] = raw[blk * kMtN + i];
        if (tid == 0) {
            v.hdr->mt_idx = idx;
            v.hdr->words += (unsigned long long)consumed;
            v.hdr->rounds_filled = job.first_round + rounds;
        }
    } else if (tail && before + total < job.need && tid == 0) {
        v.hdr->underrun = 1;     // the margin of extra words did not cover the rejections: reported, never silent
    }
}

// ---------------------------------------------------------------------------------------------
// send: Serial.send (vector.py:137-156), one thread per env
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) squared_send_kernel(SquaredView v, const long long *actions, float *obs,
                                                          float *rewards, uint8_t *terminals, uint8_t *truncations,
                                                          uint8_t *masks) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= v.n) return;
    SquaredEnv s;
    squared_load(v, e, s);
    float *grid = obs + (size_t)e * v.stride;
    uint16_t tc[kMaxTargets];
    float reward;
    bool terminal, finished = false;
    double fr = 0.0, fs = 0.0;
    int fl = 0;
    if (s.done) {
        if ((long long)s.rounds >= v.hdr->rounds_filled) v.hdr->underrun = 1;
        const uint16_t *tr = v.tape + (size_t)(s.rounds % (uint32_t)v.tape_rounds) * v.nt * v.n;
        squared_reset(v, e, s, grid, tr, tc, reward, terminal);
        for (int t = 0; t < v.nt; ++t) v.tgt[(size_t)t * v.n + e] = tc[t];
        s.rounds += 1;
    } else {
        for (int t = 0; t < v.nt; ++t) tc[t] = v.tgt[(size_t)t * v.n + e];
        squared_step(v, s, grid, tc, (int)actions[e], reward, terminal, finished, fr, fl, fs);
    }
    squared_store(v, e, s);
    v.fin[e] = finished ? 1 : 0;
    if (finished) {
        v.fin_ret[e] = fr;
        v.fin_len[e] = fl;
        v.fin_score[e] = fs;
    }
    rewards[e] = reward;
    terminals[e] = terminal ? 1 : 0;
    truncations[e] = 0;
    masks[e] = 1;
}

// deterministic fixed-order reduction of the per-env episode accumulators
__global__ void __launch_bounds__(256) squared_stats_kernel(SquaredView v, double *out4, int reset) {
    __shared__ double sh[4][256];
    // thread t sums envs t, t+256, ... in that order (the order is part of the result: f64 sums of python floats).  Whole batches of
    // eight envs issue their loads together before their stores — one memory round trip per eight envs instead of one per env (the
    // resets may alias the loads as far as the compiler knows, which serialises the plain loop); the tail runs the plain loop.
    double a[4] = {0, 0, 0, 0};
    constexpr int kBatch = 8;
    int e0 = threadIdx.x;
    for (; e0 + 256 * (kBatch - 1) < v.n; e0 += 256 * kBatch) {
        int cnt[kBatch];
        long long len[kBatch];
        double ret[kBatch], score[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int e = e0 + 256 * k;
            cnt[k] = v.acc_cnt[e];
            ret[k] = v.acc_ret[e];
            len[k] = v.acc_len[e];
            score[k] = v.acc_score[e];
        }
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int e = e0 + 256 * k;
            a[0] += (double)cnt[k];
            a[1] += ret[k];
            a[2] += (double)len[k];
            a[3] += score[k];
            if (reset) {
                v.acc_cnt[e] = 0;
                v.acc_ret[e] = 0.0;
                v.acc_len[e] = 0;
                v.acc_score[e] = 0.0;
            }
        }
    }
    for (int e = e0; e < v.n; e += 256) {
        a[0] += (double)v.acc_cnt[e];
        a[1] += v.acc_ret[e];
        a[2] += (double)v.acc_len[e];
        a[3] += v.acc_score[e];
        if (reset) {
            v.acc_cnt[e] = 0;
            v.acc_ret[e] = 0.0;
            v.acc_len[e] = 0;
            v.acc_score[e] = 0.0;
        }
    }
    for (int q = 0; q < 4; ++q) sh[q][threadIdx.x] = a[q];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int q = 0; q < 4; ++q) sh[q][threadIdx.x] += sh[q][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 4) out4[threadIdx.x] = sh[threadIdx.x][0];
    if (threadIdx.x == 4) out4[4] = (double)v.hdr->underrun;   // a reset found no tape round: host bookkeeping error, never silent
}

__global__ void squared_infos_kernel(SquaredView v, uint8_t *fin, double *ret, int32_t *len, double *score) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= v.n) return;
    fin[e] = v.fin[e];
    ret[e] = v.fin_ret[e];
    len[e] = v.fin_len[e];
    score[e] = v.fin_score[e];
}

__global__ void squared_debug_targets_kernel(SquaredView v, int32_t *cells) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= v.n) return;
    const uint32_t rem = v.rem[e];
    for (int t = 0; t < v.nt; ++t) cells[(size_t)e * v.nt + t] = ((rem >> t) & 1u) ? (int)v.tgt[(size_t)t * v.n + e] : -1;
}

__global__ void squared_debug_pos_kernel(SquaredView v, unsigned long long *pos) { *pos = v.hdr->words; }

static int check_cfg(const pfa_squared_config *c) {
    PFA_REQUIRE(c != nullptr, "squared: null config");
    PFA_REQUIRE(c->num_envs >= 1, "squared: num_envs must be at least 1");
    PFA_REQUIRE(c->distance_to_target >= 1 && c->distance_to_target <= 15, "squared: distance_to_target must be in 1..15");
    PFA_REQUIRE(c->num_targets >= 1 && c->num_targets <= kMaxTargets && c->num_targets <= 8 * c->distance_to_target,
                "squared: num_targets must be in 1..min(32, 8*distance_to_target)");
    const int g = 2 * c->distance_to_target + 1;
    PFA_REQUIRE(c->obs_stride >= g * g, "squared: obs_stride %d < grid cells %d", c->obs_stride, g * g);
    PFA_REQUIRE(c->tape_rounds >= 1, "squared: tape_rounds must be >= 1");
    return 0;
}

}  // namespace pfa

using namespace pfa;

extern "C" size_t pfa_squared_state_bytes(const pfa_squared_config *cfg) {
    if (check_cfg(cfg)) return 0;
    size_t total = 0;
    squared_view(nullptr, *cfg, &total);
    return total;
}

extern "C" int pfa_squared_async_reset(void *state, const pfa_squared_config *cfg, int64_t seed, float *obs, float *rewards,
                                       uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    if (int rc = check_cfg(cfg)) return rc;
    PFA_REQUIRE(state && obs && rewards && terminals && truncations && masks, "squared.async_reset: null buffer");
    SquaredView v = squared_view(state, *cfg);
    const int blocks = (cfg->num_envs + 255) / 256;
    hipLaunchKernelGGL(squared_seed_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, v, (long long)seed, obs, rewards,
                       terminals, truncations, masks);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_squared_fill_tape(void *state, const pfa_squared_config *cfg, int32_t rounds, pfa_stream_t stream) {
    if (int rc = check_cfg(cfg)) return rc;
    PFA_REQUIRE(state, "squared.fill_tape: null state");
    PFA_REQUIRE(rounds >= 0 && rounds <= cfg->tape_rounds, "squared.fill_tape: rounds %d exceeds tape capacity %d", rounds,
                cfg->tape_rounds);
    if (rounds == 0) return 0;
    SquaredView v = squared_view(state, *cfg);
    ScopedKernelTimer timer("squared_tape", (hipStream_t)stream);
    if (cfg->num_targets == 1) {
        // single-target form: one small workgroup of raw words + two parallel passes (see above); needs room in the seeding scratch
        const int n_pop = 8 * cfg->distance_to_target;
        int bits = 0;
        while ((1 << bits) <= n_pop) ++bits;                       // n_pop.bit_length()
        const double need = (double)rounds * cfg->num_envs;
        const double words = need * (double)(1 << bits) / n_pop * 1.02 + 6.0 * kMtN;
        const long long blocks = (long long)(words / kMtN) + 1;
        const long long chunks = ((blocks + 1) * kMtN + kSelChunk - 1) / kSelChunk;
        const long long scratch_words = (long long)kMtN * cfg->num_envs;
        const long long used = (blocks + 1) * kMtN + 2 * kMtStep + chunks + 64;    // raw words (+ one super-step of slack), counts, job
        if (used <= scratch_words && blocks < (1 << 21)) {
            uint32_t *raw = v.seed_mt;
            int *counts = (int *)(raw + (blocks + 1) * kMtN + 2 * kMtStep);
            TapeJob *job = (TapeJob *)(((uintptr_t)(counts + chunks) + 15) & ~(uintptr_t)15);
            hipLaunchKernelGGL(squared_tape_words_kernel, dim3(1), dim3(kTapeWordThreads), 0, (hipStream_t)stream, v, raw, job, (int)rounds, (int)blocks);
            hipLaunchKernelGGL(squared_tape_count_kernel, dim3((unsigned)chunks), dim3(kSelThreads), 0, (hipStream_t)stream, v,
                               (const uint32_t *)raw, (const TapeJob *)job, counts);
            hipLaunchKernelGGL(squared_tape_select_kernel, dim3((unsigned)chunks), dim3(kSelThreads), 0, (hipStream_t)stream, v,
                               (const uint32_t *)raw, (const TapeJob *)job, (const int *)counts, (int)rounds);
            PFA_LAUNCH_CHECK();
            return 0;
        }
    }
    hipLaunchKernelGGL(squared_tape_kernel, dim3(1), dim3(kTapeThreads), 0, (hipStream_t)stream, v, (int)rounds);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_squared_send(void *state, const pfa_squared_config *cfg, const int64_t *actions, float *obs, float *rewards,
                                uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    if (int rc = check_cfg(cfg)) return rc;
    PFA_REQUIRE(state && actions && obs && rewards && terminals && truncations && masks, "squared.send: null buffer");
    SquaredView v = squared_view(state, *cfg);
    const int blocks = (cfg->num_envs + 255) / 256;
    hipLaunchKernelGGL(squared_send_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, v, (const long long *)actions, obs,
                       rewards, terminals, truncations, masks);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_squared_episode_stats(void *state, const pfa_squared_config *cfg, double *out4, int32_t reset,
                                         pfa_stream_t stream) {
    if (int rc = check_cfg(cfg)) return rc;
    PFA_REQUIRE(state && out4, "squared.episode_stats: null buffer");
    SquaredView v = squared_view(state, *cfg);
    hipLaunchKernelGGL(squared_stats_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, v, out4, (int)reset);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_squared_last_infos(void *state, const pfa_squared_config *cfg, uint8_t *finished, double *episode_return,
                                      int32_t *episode_length, double *score, pfa_stream_t stream) {
    if (int rc = check_cfg(cfg)) return rc;
    PFA_REQUIRE(state && finished && episode_return && episode_length && score, "squared.last_infos: null buffer");
    SquaredView v = squared_view(state, *cfg);
    const int blocks = (cfg->num_envs + 255) / 256;
    hipLaunchKernelGGL(squared_infos_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, v, finished, episode_return,
                       episode_length, score);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_squared_debug_targets(void *state, const pfa_squared_config *cfg, int32_t *cells, pfa_stream_t stream) {
    if (int rc = check_cfg(cfg)) return rc;
    SquaredView v = squared_view(state, *cfg);
    const int blocks = (cfg->num_envs + 255) / 256;
    hipLaunchKernelGGL(squared_debug_targets_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, v, cells);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_squared_debug_stream_pos(void *state, const pfa_squared_config *cfg, uint64_t *pos_device, pfa_stream_t stream) {
    if (int rc = check_cfg(cfg)) return rc;
    SquaredView v = squared_view(state, *cfg);
    hipLaunchKernelGGL(squared_debug_pos_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, v, (unsigned long long *)pos_device);
    PFA_LAUNCH_CHECK();
    return 0;
}
