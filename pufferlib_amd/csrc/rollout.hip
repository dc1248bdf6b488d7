// rollout.hip — policy forward + action sampling, standalone (frameworks/cleanrl.py:60-66 Policy.forward with
// action=None) and fused with the env into a persistent rollout (clean_pufferl.py:76-154 evaluate).
//
// Fused rollout: one wavefront owns 16 envs for all T steps.  Env state sits in registers of lanes 0..15,
// the 16 observation grids sit in LDS (they double as the MFMA B operand), the policy weights sit in
// registers as MFMA A fragments for the whole launch.  Nothing is exchanged between workgroups: the only
// cross-env coupling in the reference — the shared `random.sample` stream — was resolved ahead of time into
// the reset-target tape (squared.hip).  Per step the HBM traffic is the experience row itself:
// obs_stride*4 + 20 B per env (280 B at obs_stride 64).
#include "common.hpp"
#include "lane_ops.hpp"
#include "mlp_tile.hpp"
#include "philox.hpp"
#include "rollout_tile.hpp"
#include "sampler.hpp"
#include "squared_env.hpp"

namespace pfa {

#ifdef PFA_PROBES   // tools/probe_rollout.py: s_memtime stamps of workgroup 0's waves over a window of steps
__device__ unsigned long long *g_rtrace = nullptr;
__device__ int g_rtrace_t0 = 0, g_rtrace_steps = 0;
#define PFA_RSTAMP(t, k)                                                                                            \
    do {                                                                                                            \
        if (g_rtrace && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (t) >= g_rtrace_t0 && (t) < g_rtrace_t0 + g_rtrace_steps) \
            g_rtrace[(((size_t)(threadIdx.x >> 6)) * g_rtrace_steps + ((t) - g_rtrace_t0)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PFA_RSTAMP(t, k) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// standalone forward + sample over `rows` observation rows
// ---------------------------------------------------------------------------------------------
template <int DP, int KS, int MW = kMW>
__global__ void __launch_bounds__(kRollThreads) mlp_forward_sample_kernel(const float *obs, long long rows, MlpView pv,
                                                                         int a, uint32_t heads, const float *noise,
                                                                         uint64_t seed, uint64_t step, long long row_offset,
                                                                         long long *actions, float *logprob, float *entropy,
                                                                         float *value) {
    __shared__ float xs[XTile<DP>::kFloats];
    __shared__ float part[kRollWaves][kPartFloats];
    SliceFrags<DP, KS, MW> w;
    w.load(pv);
    const int le = threadIdx.x >> 4, lo = threadIdx.x & 15;
    const long long tiles = (rows + 15) / 16;
    for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        __syncthreads();
        stage_rows<DP>(obs, tile * 16, rows, xs);
        __syncthreads();
        forward_slice<DP, KS, MW>(w, xs, part);
        __syncthreads();
        const long long row = tile * 16 + le;
        const bool ok = row < rows;
        const float q = ok ? noise_lane(noise ? noise + row * a : nullptr, seed, step, (uint64_t)(row_offset + row), lo, a) : 1.0f;
        const LaneSample sm = heads ? sample_lanes_heads(part, le, lo, a, heads, q) : sample_lanes(part, le, lo, a, q);
        if (ok && lo == 0) {
            actions[row] = sm.action;
            logprob[row] = sm.logprob;
            if (entropy) entropy[row] = sm.entropy;
            value[row] = sm.value;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fused persistent rollout
//
// One workgroup of 4 wavefronts owns 16 envs for all T steps.  Per step:
//   all 256 threads   store the 16 observation rows (LDS -> experience, one float4 each at obs_stride 64)
//   wave w            forward_slice: 2 x KS MFMAs against W1 fragments held in registers + its 8 head MFMAs
//   barrier
//   thread (e, o)     sample_lanes: softmax / Exp(1) noise / argmax(p/q) / logprob over the row's 16 lanes
//   thread (e, 0)     Experience.store of the scalars, then send(): env step or tape reset, rewriting row e of the
//                     LDS observation tile
//   barrier
// The chain env -> forward -> sample -> env is latency-bound (one wave per SIMD, 128 dependent steps), so everything that
// does not depend on the previous step is taken off it:
//   * the Exp(1) noise of step t+1 is loaded into a register while step t runs when the caller hands the rollout a noise
//     tensor (clean_pufferl.evaluate draws the whole rollout's Philox noise with pfa_philox_exp_noise in one launch: 40
//     quarter-rate integer multiplies per step and wave otherwise);
//   * NT1 (one target, every ocean default): the next reset's target is fetched from the tape right after the previous
//     reset (a dependent HBM round trip every episode otherwise), the tape's fill level is read once at launch, and the env
//     step is the single-target form of squared_env.hpp (reward table, no integer or f64 division, two-cell grid clear);
//   * no full drain of the vector-memory counter inside the loop: the weight fragments loaded before the loop used to be "in
//     flight" at the loop header as far as the compiler could tell, so the first wait of every step was a `vmcnt(0)` — which
//     also waits for the acknowledgement of the observation-row stores issued a few instructions earlier and of the previous
//     step's scalar stores.  One explicit drain before the loop removes it; the two per-step barriers order LDS only
//     (`lds_barrier`, common.hpp).
// ---------------------------------------------------------------------------------------------
template <int DP, int EPW, bool NT1, int KS, int MW = kMW>
__global__ void __launch_bounds__(kRollThreads) rollout_mlp_squared_kernel(SquaredView v, MlpView pv, int a,
                                                                          pfa_experience ex, const float *noise,
                                                                          uint64_t seed, uint64_t step0, long long env_offset,
                                                                          float *live_obs, float *live_rew, uint8_t *live_term,
                                                                          uint8_t *live_trunc, uint8_t *live_mask) {
    constexpr int XS = XTile<DP>::XS;
    __shared__ float xs[XTile<DP>::kFloats];
    __shared__ float part[kRollWaves][kPartFloats];  // partial out^T[o][row] per wave
    __shared__ uint16_t tg[NT1 ? 1 : 16 * kMaxTargets];
    // One wave per SIMD on a dependent chain: whatever shares the CU (the reset-tape workgroup drawn on the side stream while
    // this kernel runs) must not win issue slots from it — the launch lasts as long as its slowest workgroup.
    __builtin_amdgcn_s_setprio(3);
    __shared__ RewardTable rtab;
    const int le = threadIdx.x >> 4, lo = threadIdx.x & 15;  // sampling role: local env, output index
    const int e = blockIdx.x * EPW + le;
    const bool env_ok = le < EPW && e < v.n;
    const bool owner = lo == 0 && env_ok;  // the thread that carries env `e`
    const int T = ex.horizon_T;

    SliceFrags<DP, KS, MW> w;
    w.load(pv);

    // recv(): the live buffers are the current observation / reward / terminal of every env
    stage_rows<DP>(live_obs, (long long)blockIdx.x * EPW, v.n, xs, EPW);
    if (NT1) rtab.build(v.d);
    SquaredEnv s;
    Target1 t1;
    float reward = 0.0f;
    bool terminal = false;
    long long filled = 0;        // NT1: tape rounds drawn when this launch started (the host guarantees they cover it)
    uint16_t next_cell = 0;      // NT1: target of this env's next reset (tape round s.rounds), fetched ahead of the reset
    uint32_t slot = 0;           // NT1: s.rounds % tape_rounds, kept incrementally
    const size_t tape_n = (size_t)v.n;
    if (owner) {
        squared_load(v, e, s);
        if (NT1) {
            t1.set(v.tgt[e], v.g);
            filled = v.hdr->rounds_filled;
            slot = s.rounds % (uint32_t)v.tape_rounds;
            next_cell = v.tape[(size_t)slot * tape_n + e];
        } else {
            for (int t = 0; t < v.nt; ++t) tg[le * kMaxTargets + t] = v.tgt[(size_t)t * v.n + e];
        }
        reward = live_rew[e];
        terminal = live_term[e] != 0;
    }
    // explicit noise rows [T][N][a]: this lane's column, one step ahead
    const bool noise_lane_ok = env_ok && lo < a;
    const float *nz = noise && noise_lane_ok ? noise + (size_t)e * a + lo : nullptr;
    const size_t nz_step = (size_t)v.n * a;
    float q_next = nz ? nz[0] : 1.0f;
    // Everything loaded so far (weight fragments, env state, first noise value) lands before the loop: a load still in flight at
    // the loop header makes the compiler's first in-loop wait a vmcnt(0), which then also waits for the experience stores of
    // every step (s_waitcnt vmcnt(0) expcnt(7) lgkmcnt(15) on gfx9).
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        PFA_RSTAMP(t, 0);
        // Experience.store of the observation rows (clean_pufferl.py:443), env-major: row (e, t) at e*T + t
        unstage_rows<DP>(xs, ex.obs + (size_t)t * DP, (long long)blockIdx.x * EPW, v.n, (size_t)T * DP, EPW);
        PFA_RSTAMP(t, 1);
        forward_slice<DP, KS, MW>(w, xs, part);
        PFA_RSTAMP(t, 2);
        lds_barrier();
        PFA_RSTAMP(t, 3);
        // (vmcnt is one in-order counter: every wait for a loaded value also waits for the stores issued before it.  The
        // prefetched noise and tape values are therefore consumed HERE and at the top of the owner block — where the youngest
        // outstanding store is a forward pass old — and this step's stores and prefetches are issued after those points.)
        const float q = noise ? q_next
                              : (env_ok ? noise_lane(nullptr, seed, step0 + t, (uint64_t)(env_offset + e), lo, a) : 1.0f);
#ifdef PFA_PROBES
        if (q == -12345.678f) PFA_RSTAMP(t, 7);   // (never true) pins the stamp behind the noise
#endif
        PFA_RSTAMP(t, 4);
        const LaneSample sm = sample_lanes(part, le, lo, a, q);
#ifdef PFA_PROBES
        if (sm.value == 12345.678f) PFA_RSTAMP(t, 7);
#endif
        PFA_RSTAMP(t, 5);
        if (owner) {
            const float stored_reward = reward;
            const bool stored_terminal = terminal;
            // send(): vector.py:144-151
            float *grid = xs + le * XS;
            if (NT1) {
                if (s.done) {
                    if ((long long)s.rounds >= filled) v.hdr->underrun = 1;
                    squared_reset_nt1(v, s, grid, t1, next_cell, reward, terminal);
                    s.rounds += 1;
                    slot = slot + 1 == (uint32_t)v.tape_rounds ? 0u : slot + 1;
                } else {
                    squared_step_nt1(v, s, grid, t1, rtab, sm.action, reward, terminal);
                }
            } else {
                uint16_t *tc = tg + le * kMaxTargets;
                if (s.done) {
                    if ((long long)s.rounds >= v.hdr->rounds_filled) v.hdr->underrun = 1;
                    const uint16_t *tr = v.tape + (size_t)(s.rounds % (uint32_t)v.tape_rounds) * v.nt * v.n;
                    squared_reset(v, e, s, grid, tr, tc, reward, terminal);
                    s.rounds += 1;
                } else {
                    bool fin;
                    double fr, fs;
                    int fl;
                    squared_step(v, s, grid, tc, sm.action, reward, terminal, fin, fr, fl, fs);
                }
            }
            // Experience.store of the scalars (clean_pufferl.py:443-449): what recv() returned for this step + the policy's outputs
            const size_t row = (size_t)e * T + t;
            ex.rewards[row] = stored_reward;
            ex.dones[row] = stored_terminal ? 1.0f : 0.0f;
            ex.actions[row] = sm.action;
            ex.logprobs[row] = sm.logprob;
            ex.values[row] = sm.value;
            // the next reset's target, re-fetched every step (a conditional fetch would make next_cell a phi whose copy waits for
            // the load at once); the value a reset consumes was requested at least one whole step earlier
            if (NT1) next_cell = v.tape[(size_t)slot * tape_n + e];
        }
        if (nz && t + 1 < T) q_next = nz[(size_t)(t + 1) * nz_step];
        PFA_RSTAMP(t, 6);
        lds_barrier();
        PFA_RSTAMP(t, 7);
    }

    // write back: env state + the live buffers the next recv() returns
    if (owner) {
        squared_store(v, e, s);
        if (NT1) {
            v.tgt[e] = t1.cell;
        } else {
            for (int t = 0; t < v.nt; ++t) v.tgt[(size_t)t * v.n + e] = tg[le * kMaxTargets + t];
        }
        v.fin[e] = 0;
        live_rew[e] = reward;
        live_term[e] = terminal ? 1 : 0;
        live_trunc[e] = 0;
        live_mask[e] = 1;
    }
    unstage_rows<DP>(xs, live_obs, (long long)blockIdx.x * EPW, v.n, (size_t)DP, EPW);
}

// The whole rollout's action noise in one launch: out[t][r][j] = the Exp(1) draw noise_lane() makes in place for
// (row row_offset + r, step key.step + t, column j) — one Philox call per four columns.
__global__ void __launch_bounds__(256) philox_exp_noise_kernel(float *out, long long steps, long long rows, int a, int groups,
                                                               uint64_t seed, uint64_t step0, long long row_offset) {
    const long long total = steps * rows * groups;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int j4 = (int)(i % groups);
        const long long tr = i / groups;
        const long long r = tr % rows, t = tr / rows;
        const uint64_t step = step0 + (uint64_t)t, row = (uint64_t)(row_offset + r);
        const u32x4 w = philox4x32_10((uint32_t)row, (uint32_t)j4, (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed,
                                      (uint32_t)(seed >> 32));
        float *o = out + (size_t)tr * a + 4 * j4;
        const int left = a - 4 * j4;
        o[0] = -logf(philox_uniform(w.x));
        if (left > 1) o[1] = -logf(philox_uniform(w.y));
        if (left > 2) o[2] = -logf(philox_uniform(w.z));
        if (left > 3) o[3] = -logf(philox_uniform(w.w));
    }
}

static int check_dims(const pfa_mlp_dims *d) {
    PFA_REQUIRE(d != nullptr, "mlp: null dims");
    PFA_REQUIRE(d->hidden == kHidden, "mlp: hidden must be %d (got %d)", kHidden, d->hidden);
    PFA_REQUIRE(d->obs_stride == 16 || d->obs_stride == 32 || d->obs_stride == 64 || d->obs_stride == 96 ||
                    d->obs_stride == 128,
                "mlp: obs_stride must be one of 16/32/64/96/128 (got %d)", d->obs_stride);
    PFA_REQUIRE(d->obs_dim >= 1 && d->obs_dim <= d->obs_stride, "mlp: obs_dim %d out of range", d->obs_dim);
    PFA_REQUIRE(d->num_actions >= 1 && d->num_actions <= 15, "mlp: num_actions must be in 1..15 (got %d)", d->num_actions);
    PFA_REQUIRE(d->heads == 0 || heads_count(d->heads, d->num_actions) >= 1, "mlp: head sizes 0x%x do not sum to num_actions %d",
                d->heads, d->num_actions);
    return 0;
}

// DP = observation row stride, KS = forward k-steps: the 7x7 grid's 49 columns need 13 of the 64-float row's 16 (rollout_tile.hpp)
#define PFA_DISPATCH_DP(dims_, CALL)                                                                     \
    if ((dims_)->obs_stride == 64 && (dims_)->obs_dim > 48 && (dims_)->obs_dim <= 52) {                  \
        constexpr int DP = 64, KS = 13; CALL;                                                            \
    } else switch ((dims_)->obs_stride) {                                                                \
        case 16: { constexpr int DP = 16, KS = 4; CALL; } break;                                          \
        case 32: { constexpr int DP = 32, KS = 8; CALL; } break;                                          \
        case 64: { constexpr int DP = 64, KS = 16; CALL; } break;                                         \
        case 96: { constexpr int DP = 96, KS = 24; CALL; } break;                                         \
        default: { constexpr int DP = 128, KS = 32; CALL; } break;                                        \
    }

}  // namespace pfa

using namespace pfa;

#ifdef PFA_PROBES
extern "C" int pfa_probe_set_rollout_trace(unsigned long long *buf, int t0, int steps) {
    PFA_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rtrace), &buf, sizeof(buf)));
    PFA_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rtrace_t0), &t0, sizeof(t0)));
    PFA_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rtrace_steps), &steps, sizeof(steps)));
    return 0;
}
#endif

extern "C" int64_t pfa_mlp_param_count(const pfa_mlp_dims *dims) {
    if (check_dims(dims)) return -1;
    return mlp_offsets(dims->obs_stride, dims->num_actions).count;
}

extern "C" int pfa_mlp_forward_sample(const float *obs, int64_t rows, const float *params, const pfa_mlp_dims *dims,
                                      const float *noise, const pfa_noise_key *key, int64_t row_offset, int64_t *actions,
                                      float *logprob, float *entropy, float *value, pfa_stream_t stream) {
    if (int rc = check_dims(dims)) return rc;
    PFA_REQUIRE(rows >= 0, "mlp.forward: negative rows");
    if (rows == 0) return 0;
    PFA_REQUIRE(obs && params && actions && logprob && value, "mlp.forward: null buffer");
    PFA_REQUIRE(noise || key, "mlp.forward: need an explicit noise tensor or a Philox key");
    const uint64_t seed = key ? key->seed : 0, step = key ? key->step : 0;
    const int64_t tiles = (rows + 15) / 16;
    const unsigned grid = (unsigned)(tiles < 4096 ? tiles : 4096);
    PFA_DISPATCH_DP(dims,
                    hipLaunchKernelGGL((mlp_forward_sample_kernel<DP, KS>), dim3(grid), dim3(kRollThreads), 0, (hipStream_t)stream, obs,
                                       (long long)rows, mlp_view_of_flat(params, DP, dims->num_actions), dims->num_actions, dims->heads, noise, seed,
                                       step, (long long)row_offset, (long long *)actions, logprob, entropy, value));
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_philox_exp_noise(float *out, int64_t steps, int64_t rows, int32_t num_actions, const pfa_noise_key *key,
                                    int64_t row_offset, pfa_stream_t stream) {
    PFA_REQUIRE(steps >= 0 && rows >= 0, "philox noise: negative shape");
    PFA_REQUIRE(num_actions >= 1 && num_actions <= 15, "philox noise: num_actions must be in 1..15 (got %d)", num_actions);
    if (steps == 0 || rows == 0) return 0;
    PFA_REQUIRE(out && key, "philox noise: null buffer");
    const int groups = (num_actions + 3) / 4;
    const long long total = (long long)steps * rows * groups;
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    ScopedKernelTimer timer("philox_exp_noise", (hipStream_t)stream);
    hipLaunchKernelGGL(philox_exp_noise_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, (long long)steps, (long long)rows,
                       (int)num_actions, groups, key->seed, key->step, (long long)row_offset);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_rollout_mlp_squared(void *state, const pfa_squared_config *cfg, const float *params,
                                       const pfa_mlp_dims *dims, const pfa_experience *exp, const float *noise,
                                       const pfa_noise_key *key, int64_t env_offset, float *obs, float *rewards,
                                       uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    if (int rc = check_dims(dims)) return rc;
    PFA_REQUIRE(state && cfg && params && exp && obs && rewards && terminals && truncations && masks, "rollout: null buffer");
    PFA_REQUIRE(dims->heads == 0, "rollout: the fused rollout samples one Discrete head");
    PFA_REQUIRE(cfg->obs_stride == dims->obs_stride, "rollout: env obs_stride %d != policy obs_stride %d", cfg->obs_stride,
                dims->obs_stride);
    PFA_REQUIRE(exp->horizon_T >= 1, "rollout: horizon must be >= 1");
    PFA_REQUIRE(exp->obs && exp->actions && exp->logprobs && exp->values && exp->rewards && exp->dones,
                "rollout: null experience buffer");
    PFA_REQUIRE(noise || key, "rollout: need an explicit noise tensor or a Philox key");
    PFA_REQUIRE(cfg->num_targets <= kMaxTargets, "rollout: too many targets");
    SquaredView v = squared_view(state, *cfg);
    const uint64_t seed = key ? key->seed : 0, step = key ? key->step : 0;
    ScopedKernelTimer timer("rollout_mlp_squared", (hipStream_t)stream);
    // 16 envs per workgroup (an 8-env variant — two workgroups per CU at N = 4096 — and four helper waves for the stores and the
    // noise both measured no better in round 2: tools/probe_rollout.py); the single-target env form when it applies.
    const unsigned grid = (unsigned)((cfg->num_envs + 15) / 16);
    if (cfg->num_targets == 1) {
        PFA_DISPATCH_DP(dims,
                        hipLaunchKernelGGL((rollout_mlp_squared_kernel<DP, 16, true, KS>), dim3(grid), dim3(kRollThreads), 0, (hipStream_t)stream,
                                           v, mlp_view_of_flat(params, DP, dims->num_actions), dims->num_actions, *exp, noise, seed, step,
                                           (long long)env_offset, obs, rewards, terminals, truncations, masks));
    } else {
        PFA_DISPATCH_DP(dims,
                        hipLaunchKernelGGL((rollout_mlp_squared_kernel<DP, 16, false, KS>), dim3(grid), dim3(kRollThreads), 0, (hipStream_t)stream,
                                           v, mlp_view_of_flat(params, DP, dims->num_actions), dims->num_actions, *exp, noise, seed, step,
                                           (long long)env_offset, obs, rewards, terminals, truncations, masks));
    }
    PFA_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// The same two kernels for a Default policy of another width (models.py:24-39: hidden_size 64 / 256 / 512 — what
// environments/classic_control, nethack / nmmo and atari's MLP heads use), reading the module's own tensors through a
// pfa_mlp_view: hidden = 64 MW with MW hidden tiles per wave, W1 fragments in registers for the whole launch like the 128-wide
// instantiation.  One Discrete head of up to 15 actions, observation rows of 16 / 32 / 64 floats.
// ---------------------------------------------------------------------------------------------
static int check_view(const pfa_mlp_view *p) {
    PFA_REQUIRE(p != nullptr, "mlp view: null");
    PFA_REQUIRE(p->w1 && p->b1 && p->w2 && p->b2 && p->wv && p->bv, "mlp view: null tensor");
    PFA_REQUIRE(p->hidden == 64 || p->hidden == 128 || p->hidden == 256 || p->hidden == 512, "mlp view: hidden %d is not one of 64/128/256/512", p->hidden);
    PFA_REQUIRE(p->obs_stride == 16 || p->obs_stride == 32 || p->obs_stride == 64, "mlp view: obs_stride %d is not one of 16/32/64", p->obs_stride);
    PFA_REQUIRE(p->obs_dim >= 1 && p->obs_dim <= p->obs_stride && p->ldw1 >= p->obs_dim, "mlp view: obs_dim %d / ldw1 %d out of range", p->obs_dim, p->ldw1);
    PFA_REQUIRE(p->num_actions >= 1 && p->num_actions <= 15, "mlp view: num_actions must be in 1..15 (got %d)", p->num_actions);
    return 0;
}
extern "C" int pfa_mlp_view_supported(const pfa_mlp_view *p) { return check_view(p) == 0 ? 1 : 0; }
static MlpView device_view(const pfa_mlp_view *p) {
    return MlpView{p->w1, p->ldw1, p->obs_dim, p->b1, p->w2, p->b2, p->wv, p->bv, p->num_actions, p->hidden};
}
// (variadic: the launch expression is macro-expanded before it reaches the inner dispatch and then holds top-level commas)
#define PFA_DISPATCH_VIEW_DP(view_, ...)                                                                 \
    if ((view_)->obs_stride == 64 && (view_)->obs_dim > 48 && (view_)->obs_dim <= 52) {                  \
        constexpr int DP = 64, KS = 13; __VA_ARGS__;                                                     \
    } else switch ((view_)->obs_stride) {                                                                \
        case 16: { constexpr int DP = 16, KS = 4; __VA_ARGS__; } break;                                   \
        case 32: { constexpr int DP = 32, KS = 8; __VA_ARGS__; } break;                                   \
        default: { constexpr int DP = 64, KS = 16; __VA_ARGS__; } break;                                  \
    }
#define PFA_DISPATCH_VIEW(view_, ...)                                                                    \
    switch ((view_)->hidden) {                                                                           \
        case 64: { constexpr int MW = 1; PFA_DISPATCH_VIEW_DP(view_, __VA_ARGS__) } break;                \
        case 128: { constexpr int MW = 2; PFA_DISPATCH_VIEW_DP(view_, __VA_ARGS__) } break;               \
        case 256: { constexpr int MW = 4; PFA_DISPATCH_VIEW_DP(view_, __VA_ARGS__) } break;               \
        default: { constexpr int MW = 8; PFA_DISPATCH_VIEW_DP(view_, __VA_ARGS__) } break;                \
    }

extern "C" int pfa_mlp_view_forward_sample(const float *obs, int64_t rows, const pfa_mlp_view *view, const float *noise,
                                           const pfa_noise_key *key, int64_t row_offset, int64_t *actions, float *logprob,
                                           float *entropy, float *value, pfa_stream_t stream) {
    if (int rc = check_view(view)) return rc;
    PFA_REQUIRE(rows >= 0, "mlp.forward: negative rows");
    if (rows == 0) return 0;
    PFA_REQUIRE(obs && actions && logprob && value, "mlp.forward: null buffer");
    PFA_REQUIRE(noise || key, "mlp.forward: need an explicit noise tensor or a Philox key");
    const uint64_t seed = key ? key->seed : 0, step = key ? key->step : 0;
    const int64_t tiles = (rows + 15) / 16;
    const unsigned grid = (unsigned)(tiles < 4096 ? tiles : 4096);
    const MlpView pv = device_view(view);
    PFA_DISPATCH_VIEW(view,
                      hipLaunchKernelGGL((mlp_forward_sample_kernel<DP, KS, MW>), dim3(grid), dim3(kRollThreads), 0, (hipStream_t)stream, obs,
                                         (long long)rows, pv, view->num_actions, 0u, noise, seed, step, (long long)row_offset,
                                         (long long *)actions, logprob, entropy, value));
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_rollout_mlp_view_squared(void *state, const pfa_squared_config *cfg, const pfa_mlp_view *view, const pfa_experience *exp,
                                            const float *noise, const pfa_noise_key *key, int64_t env_offset, float *obs, float *rewards,
                                            uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    if (int rc = check_view(view)) return rc;
    PFA_REQUIRE(state && cfg && exp && obs && rewards && terminals && truncations && masks, "rollout: null buffer");
    PFA_REQUIRE(cfg->obs_stride == view->obs_stride, "rollout: env obs_stride %d != policy obs_stride %d", cfg->obs_stride, view->obs_stride);
    PFA_REQUIRE(exp->horizon_T >= 1, "rollout: horizon must be >= 1");
    PFA_REQUIRE(exp->obs && exp->actions && exp->logprobs && exp->values && exp->rewards && exp->dones, "rollout: null experience buffer");
    PFA_REQUIRE(noise || key, "rollout: need an explicit noise tensor or a Philox key");
    PFA_REQUIRE(cfg->num_targets <= kMaxTargets, "rollout: too many targets");
    SquaredView v = squared_view(state, *cfg);
    const uint64_t seed = key ? key->seed : 0, step = key ? key->step : 0;
    const MlpView pv = device_view(view);
    ScopedKernelTimer timer("rollout_mlp_squared", (hipStream_t)stream);
    const unsigned grid = (unsigned)((cfg->num_envs + 15) / 16);
    if (cfg->num_targets == 1) {
        PFA_DISPATCH_VIEW(view,
                          hipLaunchKernelGGL((rollout_mlp_squared_kernel<DP, 16, true, KS, MW>), dim3(grid), dim3(kRollThreads), 0, (hipStream_t)stream,
                                             v, pv, view->num_actions, *exp, noise, seed, step, (long long)env_offset, obs, rewards, terminals,
                                             truncations, masks));
    } else {
        PFA_DISPATCH_VIEW(view,
                          hipLaunchKernelGGL((rollout_mlp_squared_kernel<DP, 16, false, KS, MW>), dim3(grid), dim3(kRollThreads), 0, (hipStream_t)stream,
                                             v, pv, view->num_actions, *exp, noise, seed, step, (long long)env_offset, obs, rewards, terminals,
                                             truncations, masks));
    }
    PFA_LAUNCH_CHECK();
    return 0;
}
