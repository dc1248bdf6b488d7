// stochastic.hip — ocean `Stochastic` (pufferlib/environments/ocean/ocean.py:529-582) as a device-resident vecenv: the second
// env family behind the same kernel shape as Squared.  What the reference runs per env is
// pufferlib.vector.Serial (vector.py:78-162) over make_stochastic (ocean/environment.py:61-64: horizon fixed to 100) =
// GymnasiumPufferEnv (emulation.py:169-228) over EpisodeStats (postprocess.py:18-54) over the env itself.
//
// The env has no randomness and no cross-env coupling: state = (tick, count of action 0), observation always [0.0],
//   reward = 1 - (p - count/tick)^2 in f64 if the action moved the action-0 fraction towards p, else 0   (ocean.py:566-576)
// cast to f32 by the buffer write (emulation.py:219), terminal when tick == horizon, the next send() is the auto-reset row
// (reward 0, terminal False; vector.py:147-151).  EpisodeStats: f64 sum of the episode's rewards in order, length, score =
// the last proximity.  Kernels: async_reset, send (protocol path), episode statistics, and the fused persistent rollout
// with the MLP policy (same tile code as every other rollout, rollout_tile.hpp; observation row stride 16).
#include "common.hpp"
#include "episode_fin.hpp"
#include "rollout_tile.hpp"

namespace pfa {

constexpr int kStoDP = 16;  // observation row stride in floats (1 real column)

struct StochasticEnv {
    int tick, count, done, ep_length;
    double ep_return;
};
using StochasticFin = EpisodeFin;  // episode_fin.hpp
struct StochasticView {
    StochasticEnv *env;
    StochasticFin *fin;
    int n;
};
__host__ __device__ inline size_t stochastic_state_bytes(int n) {
    return (size_t)n * (sizeof(StochasticEnv) + sizeof(StochasticFin));
}
__host__ __device__ inline StochasticView stochastic_view(void *state, int n) {
    StochasticView v;
    v.env = (StochasticEnv *)state;
    v.fin = (StochasticFin *)((char *)state + (size_t)n * sizeof(StochasticEnv));
    v.n = n;
    return v;
}

__device__ __forceinline__ void stochastic_reset(StochasticEnv &s, float &reward, bool &terminal) {
    s.tick = s.count = 0;
    s.done = 0;
    s.ep_length = 0;
    s.ep_return = 0.0;
    reward = 0.0f;
    terminal = false;
}

// One env step (ocean.py:562-582 + postprocess.py:22-54 + emulation.py:194-228).  Returns true when the episode finished.
__device__ __forceinline__ bool stochastic_step(StochasticEnv &s, int action, double p, int horizon, float &reward, bool &terminal,
                                                double &fin_return, int &fin_length, double &fin_score) {
#pragma clang fp contract(off)  // python evaluates 1 - (p - frac)**2 with one rounding per operation: no fma contraction here
    s.tick += 1;
    s.count += action == 0;
    const double frac = (double)s.count / (double)s.tick;
    const double diff = p - frac;
    const double proximity = 1.0 - diff * diff;
    const double r = ((action == 0 && frac < p) || (action == 1 && frac >= p)) ? proximity : 0.0;
    s.ep_return += r;
    s.ep_length += 1;
    reward = (float)r;
    terminal = s.tick == horizon;
    s.done = terminal;
    if (terminal) {
        fin_return = s.ep_return;
        fin_length = s.ep_length;
        fin_score = proximity;
    }
    return terminal;
}

__global__ void __launch_bounds__(256) stochastic_reset_kernel(StochasticView v, float *obs, float *rewards, uint8_t *terminals,
                                                              uint8_t *truncations, uint8_t *masks) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= v.n) return;
    StochasticEnv s;
    float r;
    bool t;
    stochastic_reset(s, r, t);
    v.env[e] = s;
    StochasticFin f = {};
    v.fin[e] = f;
#pragma unroll
    for (int k = 0; k < kStoDP; ++k) obs[(size_t)e * kStoDP + k] = 0.0f;
    rewards[e] = 0.0f;
    terminals[e] = 0;
    truncations[e] = 0;
    masks[e] = 1;
}

__global__ void __launch_bounds__(256) stochastic_send_kernel(StochasticView v, double p, int horizon, const long long *actions,
                                                             float *obs, float *rewards, uint8_t *terminals, uint8_t *truncations,
                                                             uint8_t *masks) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= v.n) return;
    StochasticEnv s = v.env[e];
    float r;
    bool t;
    v.fin[e].last_fin = 0;
    if (s.done) {
        stochastic_reset(s, r, t);
    } else {
        double fr, fs;
        int fl;
        if (stochastic_step(s, (int)actions[e], p, horizon, r, t, fr, fl, fs)) episode_account(v.fin[e], fr, fl, fs);
    }
    v.env[e] = s;
    obs[(size_t)e * kStoDP] = 0.0f;
    rewards[e] = r;
    terminals[e] = t ? 1 : 0;
    truncations[e] = 0;
    masks[e] = 1;
}

// Fused persistent rollout, structure of rollout_mlp_squared_kernel (rollout.hip): 16 envs per 4-wave workgroup, env state
// in registers of the owner lanes, the 16 observation rows in LDS as the MFMA B operand, weights in registers.
__global__ void __launch_bounds__(kRollThreads) rollout_mlp_stochastic_kernel(StochasticView v, double p, int horizon_env,
                                                                             const float *params, int a, pfa_experience ex,
                                                                             const float *noise, uint64_t seed, uint64_t step0,
                                                                             long long env_offset, float *live_obs, float *live_rew,
                                                                             uint8_t *live_term, uint8_t *live_trunc,
                                                                             uint8_t *live_mask) {
    constexpr int DP = kStoDP;
    __shared__ float xs[XTile<DP>::kFloats];
    __shared__ float part[kRollWaves][kPartFloats];
    const int le = threadIdx.x >> 4, lo = threadIdx.x & 15;
    const int e = blockIdx.x * 16 + le;
    const bool env_ok = e < v.n;
    const bool owner = lo == 0 && env_ok;
    const int T = ex.horizon_T;

    SliceFrags<DP> w;
    w.load(params, a);
    stage_rows<DP>(live_obs, (long long)blockIdx.x * 16, v.n, xs, 16);
    StochasticEnv s = {};
    float reward = 0.0f;
    bool terminal = false;
    if (owner) {
        s = v.env[e];
        reward = live_rew[e];
        terminal = live_term[e] != 0;
        v.fin[e].last_fin = 0;
    }
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        unstage_rows<DP>(xs, ex.obs + (size_t)t * DP, (long long)blockIdx.x * 16, v.n, (size_t)T * DP, 16);
        forward_slice<DP>(w, xs, part);
        __syncthreads();
        const float q = env_ok ? noise_lane(noise ? noise + ((size_t)t * v.n + e) * a : nullptr, seed, step0 + t,
                                            (uint64_t)(env_offset + e), lo, a)
                               : 1.0f;
        const LaneSample sm = sample_lanes(part, le, lo, a, q);
        if (owner) {
            const size_t row = (size_t)e * T + t;
            ex.rewards[row] = reward;
            ex.dones[row] = terminal ? 1.0f : 0.0f;
            ex.actions[row] = sm.action;
            ex.logprobs[row] = sm.logprob;
            ex.values[row] = sm.value;
            if (s.done) {
                stochastic_reset(s, reward, terminal);
            } else {
                double fr, fs;
                int fl;
                if (stochastic_step(s, sm.action, p, horizon_env, reward, terminal, fr, fl, fs)) episode_account(v.fin[e], fr, fl, fs);
            }
            // the observation never changes: xs row le stays [0, 0, ...]
        }
        __syncthreads();
    }
    if (owner) {
        v.env[e] = s;
        live_rew[e] = reward;
        live_term[e] = terminal ? 1 : 0;
        live_trunc[e] = 0;
        live_mask[e] = 1;
    }
    unstage_rows<DP>(xs, live_obs, (long long)blockIdx.x * 16, v.n, (size_t)DP, 16);
}

}  // namespace pfa

using namespace pfa;

extern "C" size_t pfa_stochastic_state_bytes(int32_t num_envs) { return num_envs > 0 ? stochastic_state_bytes(num_envs) : 0; }

extern "C" int pfa_stochastic_async_reset(void *state, int32_t num_envs, float *obs, float *rewards, uint8_t *terminals,
                                          uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    PFA_REQUIRE(state && num_envs >= 1 && obs && rewards && terminals && truncations && masks, "stochastic.async_reset: bad arguments");
    hipLaunchKernelGGL(stochastic_reset_kernel, dim3((unsigned)((num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       stochastic_view(state, num_envs), obs, rewards, terminals, truncations, masks);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_stochastic_send(void *state, int32_t num_envs, double p, int32_t horizon, const int64_t *actions, float *obs,
                                   float *rewards, uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    PFA_REQUIRE(state && num_envs >= 1 && horizon >= 1 && actions && obs && rewards && terminals && truncations && masks,
                "stochastic.send: bad arguments");
    hipLaunchKernelGGL(stochastic_send_kernel, dim3((unsigned)((num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       stochastic_view(state, num_envs), p, (int)horizon, (const long long *)actions, obs, rewards, terminals,
                       truncations, masks);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_stochastic_episode_stats(void *state, int32_t num_envs, double *out4, int32_t reset, pfa_stream_t stream) {
    PFA_REQUIRE(state && num_envs >= 1 && out4, "stochastic.episode_stats: bad arguments");
    hipLaunchKernelGGL(episode_stats_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, stochastic_view(state, num_envs).fin,
                       (int)num_envs, out4, (int)reset);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_stochastic_last_infos(void *state, int32_t num_envs, uint8_t *finished, double *episode_return,
                                         int32_t *episode_length, double *score, pfa_stream_t stream) {
    PFA_REQUIRE(state && num_envs >= 1 && finished && episode_return && episode_length && score, "stochastic.last_infos: bad arguments");
    hipLaunchKernelGGL(episode_infos_kernel, dim3((unsigned)((num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       stochastic_view(state, num_envs).fin, (int)num_envs, finished, episode_return, (int *)episode_length, score);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_rollout_mlp_stochastic(void *state, int32_t num_envs, double p, int32_t horizon, const float *params,
                                          const pfa_mlp_dims *dims, const pfa_experience *exp, const float *noise,
                                          const pfa_noise_key *key, int64_t env_offset, float *obs, float *rewards,
                                          uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    PFA_REQUIRE(state && num_envs >= 1 && horizon >= 1 && params && dims && exp && obs && rewards && terminals && truncations && masks,
                "rollout_stochastic: bad arguments");
    PFA_REQUIRE(dims->heads == 0, "rollout_stochastic: the fused rollout samples one Discrete head");
    PFA_REQUIRE(dims->hidden == kHidden && dims->obs_stride == kStoDP && dims->obs_dim == 1,
                "rollout_stochastic: the policy must take 1 observation value in rows of 16 floats");
    PFA_REQUIRE(dims->num_actions >= 2 && dims->num_actions <= 15, "rollout_stochastic: num_actions out of range");
    PFA_REQUIRE(exp->horizon_T >= 1 && exp->obs && exp->actions && exp->logprobs && exp->values && exp->rewards && exp->dones,
                "rollout_stochastic: null experience buffer");
    PFA_REQUIRE(noise || key, "rollout_stochastic: need an explicit noise tensor or a Philox key");
    const uint64_t seed = key ? key->seed : 0, step = key ? key->step : 0;
    ScopedKernelTimer timer("rollout_mlp_stochastic", (hipStream_t)stream);
    hipLaunchKernelGGL(rollout_mlp_stochastic_kernel, dim3((unsigned)((num_envs + 15) / 16)), dim3(kRollThreads), 0, (hipStream_t)stream,
                       stochastic_view(state, num_envs), p, (int)horizon, params, dims->num_actions, *exp, noise, seed, step,
                       (long long)env_offset, obs, rewards, terminals, truncations, masks);
    PFA_LAUNCH_CHECK();
    return 0;
}
