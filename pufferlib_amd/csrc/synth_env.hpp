// synth_env.hpp — the synthetic byte-row env behind BASELINE configs[2] / SURVEY config C3 ("minigrid env, 4096 envs, LSTM
// policy"): MiniGrid itself is a third-party simulator that is not in the reference tree (parity unpinned), so its
// SHAPE is what the workload keeps — 160-byte uint8 observation rows (aligned struct direction:int64 + image:uint8[7,7,3],
// which models.Default reads as 160 floats, models.py:50), 7 actions, episodes cut at 100 steps (MiniGridWrapper,
// minigrid/environment.py:40-48) — filled by a counter-based generator: byte j of the row an env shows at (episode, tick) is
//   Philox4x32-10(counter = (global env, j / 16, episode, tick), key = (seed, 'SY'))  word (j % 16) / 4, byte j % 4,  mod (high + 1)
// and the reward is 1 when the action equals byte 0 of the observation the policy was shown, modulo the action count (so the
// task is learnable and stateless).  Serial's protocol (vector.py:137-156): the send after a terminal row is the reset row.
#pragma once
#include "common.hpp"
#include "episode_fin.hpp"
#include "philox.hpp"

namespace pfa {

constexpr int kSynthMaxValues = 160;

struct SynthEnv {
    int tick, done, episode, ep_correct;
    double ep_return;
};
struct SynthView {
    SynthEnv *env;
    EpisodeFin *fin;
    int n, values, stride, actions, episode_length, high;
    unsigned long long seed;
    long long env_offset;
};
__host__ __device__ inline size_t synth_state_bytes(int n) { return (size_t)n * (sizeof(SynthEnv) + sizeof(EpisodeFin)); }
__host__ __device__ inline SynthView synth_view(void *state, const pfa_synth_config &c) {
    SynthView v;
    v.env = (SynthEnv *)state;
    v.fin = (EpisodeFin *)((char *)state + (size_t)c.num_envs * sizeof(SynthEnv));
    v.n = c.num_envs;
    v.values = c.obs_values;
    v.stride = c.obs_stride;
    v.actions = c.num_actions;
    v.episode_length = c.episode_length;
    v.high = c.obs_high;
    v.seed = c.seed;
    v.env_offset = c.env_offset;
    return v;
}

// 16 consecutive row values (chunk = j / 16) of env `e` at (episode, tick), as floats
__device__ __forceinline__ void synth_chunk(const SynthView &v, int e, int episode, int tick, int chunk, float (&out)[16]) {
    const u32x4 w = philox4x32_10((uint32_t)(v.env_offset + e), (uint32_t)chunk, (uint32_t)episode, (uint32_t)tick,
                                  (uint32_t)v.seed, 0x5359u ^ (uint32_t)(v.seed >> 32));
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
    const uint32_t m = (uint32_t)v.high + 1u;
#pragma unroll
    for (int k = 0; k < 16; ++k) out[k] = (float)(((ws[k >> 2] >> (8 * (k & 3))) & 0xFFu) % m);
}

// one env step given the action and byte 0 of the observation it answers; returns true when the episode finished
__device__ __forceinline__ bool synth_step(const SynthView &v, SynthEnv &s, int action, int shown0, float &reward, bool &terminal,
                                           double &fin_return, int &fin_length, double &fin_score) {
    const int correct = action == shown0 % v.actions;
    s.tick += 1;
    s.ep_correct += correct;
    s.ep_return += (double)correct;
    reward = (float)correct;
    terminal = s.tick >= v.episode_length;
    s.done = terminal;
    if (terminal) {
        fin_return = s.ep_return;
        fin_length = s.tick;
        fin_score = (double)s.ep_correct / (double)s.tick;
    }
    return terminal;
}
__device__ __forceinline__ void synth_begin_episode(SynthEnv &s, float &reward, bool &terminal) {
    s.tick = 0;
    s.done = 0;
    s.episode += 1;
    s.ep_correct = 0;
    s.ep_return = 0.0;
    reward = 0.0f;
    terminal = false;
}

}  // namespace pfa
