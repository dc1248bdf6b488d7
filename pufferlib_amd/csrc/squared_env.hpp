// squared_env.hpp — device-side ocean Squared (pufferlib/environments/ocean/ocean.py:406-513) wrapped the way
// the reference wraps it: GymnasiumPufferEnv buffer writes (pufferlib/emulation.py:169-228), EpisodeStats
// (pufferlib/postprocess.py:18-54) and Serial's auto-reset rule (pufferlib/vector.py:137-156).
//
// HBM layout of the env state block (one allocation, SoA, N = num_envs, NT = num_targets):
//   Header            MT19937 state of the process-global `random` generator + tape cursors
//   ax, ay  i8 [N]    agent position               tick i32 [N]      done u8 [N]
//   rem     u32[N]    bitmask of targets not yet hit (bit t <-> tgt[t])
//   tgt     u16[NT][N] target cells (x*g+y) of the current episode
//   rounds  u32[N]    how many unseeded resets this env has done (= tape round it reads next)
//   ep_ret  f64[N], ep_len i32[N]                 running EpisodeStats of the current episode
//   fin     u8 [N], fin_ret f64[N], fin_len i32[N], fin_score f64[N]   infos of the last send
//   acc_cnt i32[N], acc_ret f64[N], acc_len i64[N], acc_score f64[N]   per-env sums since the last stats reset
//   tape    u16[tape_rounds][NT][N]   pre-drawn target cells per reset round (ring)
//   seed_mt u32[624][N]               scratch: one MT19937 state per env during async_reset
#pragma once
#include "common.hpp"
#include "mt19937.hpp"

namespace pfa {

constexpr int kMaxTargets = 32;

struct SquaredHeader {
    uint32_t mt[kMtN];          // shared stream state (after the last async_reset / fill_tape)
    int32_t mt_idx;             // next word to hand out (624 = regenerate first)
    int32_t underrun;           // set when a reset needed a tape round that was not filled yet
    unsigned long long words;   // words consumed since the last seeding (debug / tests)
    long long rounds_filled;    // absolute number of reset rounds drawn into the tape so far
    double stats[4];
};

struct SquaredView {
    SquaredHeader *hdr;
    int8_t *ax, *ay;
    int32_t *tick;
    uint8_t *done;
    uint32_t *rem;
    uint16_t *tgt;
    uint32_t *rounds;
    double *ep_ret;
    int32_t *ep_len;
    uint8_t *fin;
    double *fin_ret;
    int32_t *fin_len;
    double *fin_score;
    int32_t *acc_cnt;
    double *acc_ret;
    long long *acc_len;
    double *acc_score;
    uint16_t *tape;
    uint32_t *seed_mt;
    int n, d, nt, g, stride, tape_rounds;
};

__host__ __device__ inline size_t sq_carve(size_t &off, size_t bytes) {
    const size_t at = off;
    off = (off + bytes + 255) / 256 * 256;
    return at;
}

__host__ __device__ inline SquaredView squared_view(void *base, const pfa_squared_config &c, size_t *total = nullptr) {
    SquaredView v;
    const size_t N = (size_t)c.num_envs, NT = (size_t)c.num_targets;
    char *p = (char *)base;
    size_t off = 0;
    v.hdr = (SquaredHeader *)(p + sq_carve(off, sizeof(SquaredHeader)));
    v.ax = (int8_t *)(p + sq_carve(off, N));
    v.ay = (int8_t *)(p + sq_carve(off, N));
    v.tick = (int32_t *)(p + sq_carve(off, 4 * N));
    v.done = (uint8_t *)(p + sq_carve(off, N));
    v.rem = (uint32_t *)(p + sq_carve(off, 4 * N));
    v.tgt = (uint16_t *)(p + sq_carve(off, 2 * N * NT));
    v.rounds = (uint32_t *)(p + sq_carve(off, 4 * N));
    v.ep_ret = (double *)(p + sq_carve(off, 8 * N));
    v.ep_len = (int32_t *)(p + sq_carve(off, 4 * N));
    v.fin = (uint8_t *)(p + sq_carve(off, N));
    v.fin_ret = (double *)(p + sq_carve(off, 8 * N));
    v.fin_len = (int32_t *)(p + sq_carve(off, 4 * N));
    v.fin_score = (double *)(p + sq_carve(off, 8 * N));
    v.acc_cnt = (int32_t *)(p + sq_carve(off, 4 * N));
    v.acc_ret = (double *)(p + sq_carve(off, 8 * N));
    v.acc_len = (long long *)(p + sq_carve(off, 8 * N));
    v.acc_score = (double *)(p + sq_carve(off, 8 * N));
    v.tape = (uint16_t *)(p + sq_carve(off, 2 * (size_t)c.tape_rounds * NT * N));
    v.seed_mt = (uint32_t *)(p + sq_carve(off, 4 * (size_t)kMtN * N));
    v.n = c.num_envs;
    v.d = c.distance_to_target;
    v.nt = c.num_targets;
    v.g = 2 * c.distance_to_target + 1;
    v.stride = c.obs_stride;
    v.tape_rounds = c.tape_rounds;
    if (total) *total = off;
    return v;
}

// Per-env working set while an env is resident in registers (rollout kernel) or between the load
// and store of one `send`.
struct SquaredEnv {
    int x, y, tick;
    uint32_t rem;
    uint32_t rounds;
    bool done;
    double ep_ret;
    int ep_len;
    // accumulated infos of finished episodes
    int acc_cnt;
    double acc_ret;
    long long acc_len;
    double acc_score;
};

__device__ __forceinline__ void squared_load(const SquaredView &v, int e, SquaredEnv &s) {
    s.x = v.ax[e];
    s.y = v.ay[e];
    s.tick = v.tick[e];
    s.rem = v.rem[e];
    s.rounds = v.rounds[e];
    s.done = v.done[e] != 0;
    s.ep_ret = v.ep_ret[e];
    s.ep_len = v.ep_len[e];
    s.acc_cnt = v.acc_cnt[e];
    s.acc_ret = v.acc_ret[e];
    s.acc_len = v.acc_len[e];
    s.acc_score = v.acc_score[e];
}

__device__ __forceinline__ void squared_store(const SquaredView &v, int e, const SquaredEnv &s) {
    v.ax[e] = (int8_t)s.x;
    v.ay[e] = (int8_t)s.y;
    v.tick[e] = s.tick;
    v.rem[e] = s.rem;
    v.rounds[e] = s.rounds;
    v.done[e] = s.done ? 1 : 0;
    v.ep_ret[e] = s.ep_ret;
    v.ep_len[e] = s.ep_len;
    v.acc_cnt[e] = s.acc_cnt;
    v.acc_ret[e] = s.acc_ret;
    v.acc_len[e] = s.acc_len;
    v.acc_score[e] = s.acc_score;
}

// MOVES, ocean.py:424: action a moves (dx, dy).
__device__ __forceinline__ void squared_move(int a, int &dx, int &dy) {
    // dx: 0,0,-1,1,1,-1,1,-1   dy: -1,1,0,0,-1,-1,1,1
    constexpr uint32_t DX = 0x0u | (0u << 2) | (3u << 4) | (1u << 6) | (1u << 8) | (3u << 10) | (1u << 12) | (3u << 14);
    constexpr uint32_t DY = 0x3u | (1u << 2) | (0u << 4) | (0u << 6) | (3u << 8) | (3u << 10) | (1u << 12) | (1u << 14);
    const int sx = (DX >> (2 * a)) & 3, sy = (DY >> (2 * a)) & 3;
    dx = sx == 3 ? -1 : sx;
    dy = sy == 3 ? -1 : sy;
}

// Squared.reset (ocean.py:448-463) with targets taken from the tape; GymnasiumPufferEnv.reset buffer
// writes (emulation.py:187-192); EpisodeStats.reset (postprocess.py:18-20).
// `grid` is this env's observation row (obs_stride floats; any address space).  `tcells` receives the
// nt target cells (caller stores them to v.tgt).
template <typename GridPtr>
__device__ __forceinline__ void squared_reset(const SquaredView &v, int e, SquaredEnv &s, GridPtr grid,
                                              const uint16_t *tape_round /* [NT][N] */, uint16_t *tcells,
                                              float &reward, bool &terminal) {
    const int g = v.g, d = v.d, cells = g * g;
    for (int i = 0; i < cells; ++i) grid[i] = 0.0f;
    grid[d * g + d] = -1.0f;
    s.x = d;
    s.y = d;
    s.tick = 0;
    for (int t = 0; t < v.nt; ++t) {
        const uint16_t c = tape_round[(size_t)t * v.n + e];
        tcells[t] = c;
        grid[c] = 1.0f;
    }
    s.rem = v.nt >= 32 ? 0xffffffffu : ((1u << v.nt) - 1u);
    s.done = false;
    s.ep_ret = 0.0;
    s.ep_len = 0;
    reward = 0.0f;
    terminal = false;
}

// Squared.step (ocean.py:465-513) + EpisodeStats.step (postprocess.py:22-54) + GymnasiumPufferEnv.step
// buffer writes (emulation.py:219-226).  tcells: the episode's target cells.
template <typename GridPtr>
__device__ __forceinline__ void squared_step(const SquaredView &v, SquaredEnv &s, GridPtr grid, const uint16_t *tcells,
                                             int action, float &reward, bool &terminal, bool &finished,
                                             double &fin_ret, int &fin_len, double &fin_score) {
    const int g = v.g, d = v.d;
    int x = s.x, y = s.y;
    grid[x * g + y] = 0.0f;
    int dx, dy;
    squared_move(action & 7, dx, dy);
    x += dx;
    y += dy;
    int min_dist = 1 << 30;
    int hit = -1;
    for (int t = 0; t < v.nt; ++t) {
        if (!((s.rem >> t) & 1u)) continue;
        const int c = tcells[t];
        const int tx = c / g, ty = c - tx * g;
        const int ddx = x > tx ? x - tx : tx - x, ddy = y > ty ? y - ty : ty - y;
        const int dist = ddx > ddy ? ddx : ddy;
        min_dist = dist < min_dist ? dist : min_dist;
        if (dist == 0 && hit < 0) hit = t;
    }
    const double r = 1.0 - (double)min_dist / (double)d;  // python float arithmetic (ocean.py:477)
    if (hit >= 0) s.rem &= ~(1u << hit);                   // targets.remove; marker stays drawn (:495-498)
    const int ox = x > d ? x - d : d - x, oy = y > d ? y - d : d - y;
    if ((ox > oy ? ox : oy) >= d) { x = d; y = d; }         // teleport home on the perimeter (:500-504)
    s.x = x;
    s.y = y;
    grid[x * g + y] = -1.0f;
    s.tick += 1;
    const bool done = s.tick >= v.nt * d;                   // max_ticks (:438,509)
    s.ep_ret += r;                                          // sum(list) left to right in f64
    s.ep_len += 1;
    finished = done;
    if (done) {
        fin_ret = s.ep_ret;
        fin_len = s.ep_len;
        fin_score = (double)(v.nt - __popc(s.rem)) / (double)v.nt;
        s.acc_cnt += 1;
        s.acc_ret += fin_ret;
        s.acc_len += fin_len;
        s.acc_score += fin_score;
    }
    reward = (float)r;  // emulation.py:221 writes a python float into a float32 buffer
    terminal = done;
    s.done = done;      // emulation.py:226
}

// ---------------------------------------------------------------------------------------------------------------------
// Single-target form (num_targets == 1, every ocean default) for the fused rollout, where the env step sits on the
// rollout's per-step critical path.  Same statements as squared_reset / squared_step above, with what is constant per
// launch or per episode taken out of the step: the target's (x, y) lives in registers (no `c / g` per step), the reward
// 1 - k/d comes from a table of the 2d+1 possible Chebyshev distances built once per launch with the same f64 division
// (`RewardTable`: bit-identical by construction), and a reset clears the two cells that can be non-zero (agent, target
// marker) instead of the whole grid.  The fused-vs-stepwise test (which runs the general form) pins the equivalence.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kMaxRewardDist = 32;   // table entries; distances beyond it (never reached: k <= 2d <= 10) take the division

struct RewardTable {
    double rd[kMaxRewardDist];
    float rf[kMaxRewardDist];
    __device__ __forceinline__ void build(int d) {   // all threads of the workgroup; caller barriers before use
        for (int k = threadIdx.x; k < kMaxRewardDist; k += blockDim.x) {
            const double r = 1.0 - (double)k / (double)d;   // ocean.py:477 in python-float arithmetic
            rd[k] = r;
            rf[k] = (float)r;                               // emulation.py:221
        }
    }
};

struct Target1 {
    int tx, ty;
    uint16_t cell;
    __device__ __forceinline__ void set(uint16_t c, int g) {
        cell = c;
        tx = c / g;
        ty = c - tx * g;
    }
};

template <typename GridPtr>
__device__ __forceinline__ void squared_reset_nt1(const SquaredView &v, SquaredEnv &s, GridPtr grid, Target1 &tg, uint16_t new_cell,
                                                  float &reward, bool &terminal) {
    const int g = v.g, d = v.d;
    grid[s.x * g + s.y] = 0.0f;       // the only cells a finished episode leaves non-zero: the agent and the target marker
    grid[tg.cell] = 0.0f;
    grid[d * g + d] = -1.0f;
    s.x = d;
    s.y = d;
    s.tick = 0;
    tg.set(new_cell, g);
    grid[new_cell] = 1.0f;
    s.rem = 1u;
    s.done = false;
    s.ep_ret = 0.0;
    s.ep_len = 0;
    reward = 0.0f;
    terminal = false;
}

template <typename GridPtr>
__device__ __forceinline__ void squared_step_nt1(const SquaredView &v, SquaredEnv &s, GridPtr grid, const Target1 &tg,
                                                 const RewardTable &rt, int action, float &reward, bool &terminal) {
    const int g = v.g, d = v.d;
    int x = s.x, y = s.y;
    grid[x * g + y] = 0.0f;
    int dx, dy;
    squared_move(action & 7, dx, dy);
    x += dx;
    y += dy;
    const int ddx = x > tg.tx ? x - tg.tx : tg.tx - x, ddy = y > tg.ty ? y - tg.ty : tg.ty - y;
    const int min_dist = s.rem ? (ddx > ddy ? ddx : ddy) : (1 << 30);
    double r;
    float rf;
    if (min_dist < kMaxRewardDist) {
        r = rt.rd[min_dist];
        rf = rt.rf[min_dist];
    } else {
        r = 1.0 - (double)min_dist / (double)d;
        rf = (float)r;
    }
    if (min_dist == 0) s.rem = 0u;                          // targets.remove; marker stays drawn (:495-498)
    const int ox = x > d ? x - d : d - x, oy = y > d ? y - d : d - y;
    if ((ox > oy ? ox : oy) >= d) { x = d; y = d; }         // teleport home on the perimeter (:500-504)
    s.x = x;
    s.y = y;
    grid[x * g + y] = -1.0f;
    s.tick += 1;
    const bool done = s.tick >= d;                          // max_ticks = num_targets * d
    s.ep_ret += r;
    s.ep_len += 1;
    if (done) {
        s.acc_cnt += 1;
        s.acc_ret += s.ep_ret;
        s.acc_len += s.ep_len;
        s.acc_score += s.rem ? 0.0 : 1.0;                   // (1 - len(targets)) / 1
    }
    reward = rf;
    terminal = done;
    s.done = done;
}

}  // namespace pfa
