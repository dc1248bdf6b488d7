// bandit.hip — ocean `Bandit` (pufferlib/environments/ocean/ocean.py:8-63) as a device-resident vecenv.
// Reference stack per env: pufferlib.vector.Serial (vector.py:78-162) over make_bandit (ocean/environment.py:33-37) =
// GymnasiumPufferEnv + EpisodeStats + ocean.Bandit.
//
// Every reset reseeds numpy's process-global generator with hard_fixed_seed = 42 and draws solution_idx =
// randint(0, num_actions); every step terminates the episode and returns ((action == solution) + randn() * reward_scale) *
// reward_scale (the noise term only if reward_noise != 0).  All envs of a Serial vecenv reset on the same send, each reseeding
// the shared generator, so the stream every step round sees is the same: env i always receives the i-th legacy gaussian after
// seed(42) + one randint.  That makes the noise a fixed per-env table; the host side draws it once with numpy's own
// RandomState (the arithmetic the reference itself calls) and this file keeps the state machine: step row / auto-reset row,
// f64 reward with one rounding per operation cast to f32, EpisodeStats.
#include "common.hpp"
#include "episode_fin.hpp"

namespace pfa {

constexpr int kBanditDP = 16;  // observation row stride in floats (1 real column, always 1.0)

struct BanditView {
    int *done;
    EpisodeFin *fin;
    int n;
};
__host__ __device__ inline size_t bandit_state_bytes(int n) { return (size_t)n * (sizeof(EpisodeFin) + 8); }
__host__ __device__ inline BanditView bandit_view(void *state, int n) {
    BanditView v;
    v.fin = (EpisodeFin *)state;
    v.done = (int *)((char *)state + (size_t)n * sizeof(EpisodeFin));
    v.n = n;
    return v;
}

__global__ void __launch_bounds__(256) bandit_reset_kernel(BanditView v, float *obs, float *rewards, uint8_t *terminals,
                                                          uint8_t *truncations, uint8_t *masks) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= v.n) return;
    v.done[e] = 0;
    EpisodeFin f = {};
    v.fin[e] = f;
#pragma unroll
    for (int k = 0; k < kBanditDP; ++k) obs[(size_t)e * kBanditDP + k] = 0.0f;
    obs[(size_t)e * kBanditDP] = 1.0f;
    rewards[e] = 0.0f;
    terminals[e] = 0;
    truncations[e] = 0;
    masks[e] = 1;
}

__global__ void __launch_bounds__(256) bandit_send_kernel(BanditView v, int solution, double scale, const double *noise,
                                                         const long long *actions, float *obs, float *rewards, uint8_t *terminals,
                                                         uint8_t *truncations, uint8_t *masks) {
#pragma clang fp contract(off)  // (correct + noise) * scale: python floats, one rounding per operation
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= v.n) return;
    v.fin[e].last_fin = 0;
    float r = 0.0f;
    int t = 0;
    if (v.done[e]) {  // vector.py:147-149: action ignored, reset row
        v.done[e] = 0;
    } else {
        const int correct = (int)actions[e] == solution;
        const double reward = ((double)correct + (noise ? noise[e] : 0.0)) * scale;
        r = (float)reward;
        t = 1;
        v.done[e] = 1;
        episode_account(v.fin[e], reward, 1, (double)correct);
    }
    obs[(size_t)e * kBanditDP] = 1.0f;
    rewards[e] = r;
    terminals[e] = (uint8_t)t;
    truncations[e] = 0;
    masks[e] = 1;
}

}  // namespace pfa

using namespace pfa;

extern "C" size_t pfa_bandit_state_bytes(int32_t num_envs) { return num_envs > 0 ? bandit_state_bytes(num_envs) : 0; }

extern "C" int pfa_bandit_async_reset(void *state, int32_t num_envs, float *obs, float *rewards, uint8_t *terminals,
                                      uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    PFA_REQUIRE(state && num_envs >= 1 && obs && rewards && terminals && truncations && masks, "bandit.async_reset: bad arguments");
    hipLaunchKernelGGL(bandit_reset_kernel, dim3((unsigned)((num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       bandit_view(state, num_envs), obs, rewards, terminals, truncations, masks);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_bandit_send(void *state, int32_t num_envs, int32_t solution, double reward_scale, const double *noise,
                               const int64_t *actions, float *obs, float *rewards, uint8_t *terminals, uint8_t *truncations,
                               uint8_t *masks, pfa_stream_t stream) {
    PFA_REQUIRE(state && num_envs >= 1 && actions && obs && rewards && terminals && truncations && masks, "bandit.send: bad arguments");
    hipLaunchKernelGGL(bandit_send_kernel, dim3((unsigned)((num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       bandit_view(state, num_envs), (int)solution, reward_scale, noise, (const long long *)actions, obs, rewards,
                       terminals, truncations, masks);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_bandit_episode_stats(void *state, int32_t num_envs, double *out4, int32_t reset, pfa_stream_t stream) {
    PFA_REQUIRE(state && num_envs >= 1 && out4, "bandit.episode_stats: bad arguments");
    hipLaunchKernelGGL(episode_stats_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, bandit_view(state, num_envs).fin, (int)num_envs,
                       out4, (int)reset);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_bandit_last_infos(void *state, int32_t num_envs, uint8_t *finished, double *episode_return,
                                     int32_t *episode_length, double *score, pfa_stream_t stream) {
    PFA_REQUIRE(state && num_envs >= 1 && finished && episode_return && episode_length && score, "bandit.last_infos: bad arguments");
    hipLaunchKernelGGL(episode_infos_kernel, dim3((unsigned)((num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       bandit_view(state, num_envs).fin, (int)num_envs, finished, episode_return, (int *)episode_length, score);
    PFA_LAUNCH_CHECK();
    return 0;
}
