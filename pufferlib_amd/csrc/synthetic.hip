// synthetic.hip — protocol kernels of the synthetic byte-row vecenv (synth_env.hpp): the device-side generator of BASELINE
// configs[2]'s workload shape.  The fused recurrent rollout over it lives in lstm_fused.hip.
#include "common.hpp"
#include "synth_env.hpp"

namespace pfa {

__device__ __forceinline__ void synth_write_row(const SynthView &v, int e, const SynthEnv &s, float *obs) {
    float *o = obs + (size_t)e * v.stride;
    for (int chunk = 0; chunk * 16 < v.values; ++chunk) {
        float vals[16];
        synth_chunk(v, e, s.episode, s.tick, chunk, vals);
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (chunk * 16 + k < v.values) o[chunk * 16 + k] = vals[k];
    }
}

__global__ void __launch_bounds__(256) synth_reset_kernel(SynthView v, float *obs, float *rewards, uint8_t *terminals, uint8_t *truncations,
                                                         uint8_t *masks) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= v.n) return;
    SynthEnv s = {};
    s.episode = -1;
    float r;
    bool t;
    synth_begin_episode(s, r, t);
    v.env[e] = s;
    EpisodeFin f = {};
    v.fin[e] = f;
    for (int j = v.values; j < v.stride; ++j) obs[(size_t)e * v.stride + j] = 0.0f;
    synth_write_row(v, e, s, obs);
    rewards[e] = 0.0f;
    terminals[e] = 0;
    truncations[e] = 0;
    masks[e] = 1;
}

__global__ void __launch_bounds__(256) synth_send_kernel(SynthView v, const long long *actions, float *obs, float *rewards,
                                                        uint8_t *terminals, uint8_t *truncations, uint8_t *masks) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= v.n) return;
    SynthEnv s = v.env[e];
    v.fin[e].last_fin = 0;
    float r;
    bool t;
    if (s.done) {
        synth_begin_episode(s, r, t);
    } else {
        double fr, fs;
        int fl;
        const int shown0 = (int)obs[(size_t)e * v.stride];
        if (synth_step(v, s, (int)actions[e], shown0, r, t, fr, fl, fs)) episode_account(v.fin[e], fr, fl, fs);
    }
    v.env[e] = s;
    synth_write_row(v, e, s, obs);
    rewards[e] = r;
    terminals[e] = t ? 1 : 0;
    truncations[e] = 0;
    masks[e] = 1;
}

// Frame rows (BASELINE configs[3] / SURVEY config C4: uint8 (framestack, 84, 84) observations, uniform 0..255): the same counter
// stream, 16 bytes per Philox call stored as they come (obs_high = 255: byte % 256 is the byte).  One workgroup per env; lane 0
// steps the episode, everyone writes the frame.  actions == nullptr is the reset.
__global__ void __launch_bounds__(256) frames_kernel(SynthView v, const long long *actions, uint8_t *obs, float *rewards, uint8_t *terminals,
                                                    uint8_t *truncations, uint8_t *masks) {
    const int e = blockIdx.x;
    __shared__ int sh_episode, sh_tick;
    uint8_t *row = obs + (size_t)e * v.stride;
    if (threadIdx.x == 0) {
        SynthEnv s;
        float r = 0.0f;
        bool t = false;
        if (actions == nullptr) {
            s = SynthEnv{};
            s.episode = -1;
            synth_begin_episode(s, r, t);
            v.fin[e] = EpisodeFin{};
        } else {
            s = v.env[e];
            v.fin[e].last_fin = 0;
            if (s.done) {
                synth_begin_episode(s, r, t);
            } else {
                double fr, fs;
                int fl;
                if (synth_step(v, s, (int)actions[e], (int)row[0], r, t, fr, fl, fs)) episode_account(v.fin[e], fr, fl, fs);
            }
        }
        v.env[e] = s;
        rewards[e] = r;
        terminals[e] = t ? 1 : 0;
        truncations[e] = 0;
        masks[e] = 1;
        sh_episode = s.episode;
        sh_tick = s.tick;
    }
    __syncthreads();          // also orders lane 0's read of row[0] before anyone overwrites it
    const int episode = sh_episode, tick = sh_tick;
    for (int chunk = threadIdx.x; chunk * 16 < v.values; chunk += 256) {
        const u32x4 w = philox4x32_10((uint32_t)(v.env_offset + e), (uint32_t)chunk, (uint32_t)episode, (uint32_t)tick, (uint32_t)v.seed,
                                      0x5359u ^ (uint32_t)(v.seed >> 32));
        *reinterpret_cast<uint4 *>(row + (size_t)chunk * 16) = make_uint4(w.x, w.y, w.z, w.w);
    }
}

int check_frames_config(const pfa_synth_config *c) {
    PFA_REQUIRE(c != nullptr, "frames: null config");
    PFA_REQUIRE(c->num_envs >= 1, "frames: num_envs must be >= 1");
    PFA_REQUIRE(c->obs_values >= 16 && c->obs_values % 16 == 0 && c->obs_stride == c->obs_values && c->obs_high == 255,
                "frames: rows are obs_values = obs_stride bytes (a multiple of 16) of uniform 0..255");
    PFA_REQUIRE(c->num_actions >= 2 && c->num_actions <= 15, "frames: num_actions must be in 2..15");
    PFA_REQUIRE(c->episode_length >= 1, "frames: bad episode_length");
    return 0;
}

int check_synth_config(const pfa_synth_config *c) {
    PFA_REQUIRE(c != nullptr, "synth: null config");
    PFA_REQUIRE(c->num_envs >= 1, "synth: num_envs must be >= 1");
    PFA_REQUIRE(c->obs_values >= 1 && c->obs_values <= kSynthMaxValues && c->obs_stride >= c->obs_values && c->obs_stride % 16 == 0,
                "synth: obs_values must be in 1..%d and obs_stride a multiple of 16 >= obs_values", kSynthMaxValues);
    PFA_REQUIRE(c->num_actions >= 2 && c->num_actions <= 15, "synth: num_actions must be in 2..15");
    PFA_REQUIRE(c->episode_length >= 1 && c->obs_high >= 0 && c->obs_high <= 255, "synth: bad episode_length / obs_high");
    return 0;
}

// state / statistics entries serve both row kinds
static int check_synth_any(const pfa_synth_config *c) {
    return (c != nullptr && c->obs_values > kSynthMaxValues) ? check_frames_config(c) : check_synth_config(c);
}

}  // namespace pfa

using namespace pfa;

extern "C" size_t pfa_synth_state_bytes(const pfa_synth_config *cfg) { return check_synth_any(cfg) ? 0 : synth_state_bytes(cfg->num_envs); }

extern "C" int pfa_synth_async_reset(void *state, const pfa_synth_config *cfg, float *obs, float *rewards, uint8_t *terminals,
                                     uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    if (int rc = check_synth_config(cfg)) return rc;
    PFA_REQUIRE(state && obs && rewards && terminals && truncations && masks, "synth.async_reset: null buffer");
    hipLaunchKernelGGL(synth_reset_kernel, dim3((unsigned)((cfg->num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       synth_view(state, *cfg), obs, rewards, terminals, truncations, masks);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_synth_send(void *state, const pfa_synth_config *cfg, const int64_t *actions, float *obs, float *rewards,
                              uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    if (int rc = check_synth_config(cfg)) return rc;
    PFA_REQUIRE(state && actions && obs && rewards && terminals && truncations && masks, "synth.send: null buffer");
    hipLaunchKernelGGL(synth_send_kernel, dim3((unsigned)((cfg->num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       synth_view(state, *cfg), (const long long *)actions, obs, rewards, terminals, truncations, masks);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_frames_async_reset(void *state, const pfa_synth_config *cfg, uint8_t *obs, float *rewards, uint8_t *terminals,
                                      uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    if (int rc = check_frames_config(cfg)) return rc;
    PFA_REQUIRE(state && obs && rewards && terminals && truncations && masks, "frames.async_reset: null buffer");
    hipLaunchKernelGGL(frames_kernel, dim3((unsigned)cfg->num_envs), dim3(256), 0, (hipStream_t)stream, synth_view(state, *cfg),
                       (const long long *)nullptr, obs, rewards, terminals, truncations, masks);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_frames_send(void *state, const pfa_synth_config *cfg, const int64_t *actions, uint8_t *obs, float *rewards,
                               uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    if (int rc = check_frames_config(cfg)) return rc;
    PFA_REQUIRE(state && actions && obs && rewards && terminals && truncations && masks, "frames.send: null buffer");
    hipLaunchKernelGGL(frames_kernel, dim3((unsigned)cfg->num_envs), dim3(256), 0, (hipStream_t)stream, synth_view(state, *cfg),
                       (const long long *)actions, obs, rewards, terminals, truncations, masks);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_synth_episode_stats(void *state, const pfa_synth_config *cfg, double *out4, int32_t reset, pfa_stream_t stream) {
    if (int rc = check_synth_any(cfg)) return rc;
    PFA_REQUIRE(state && out4, "synth.episode_stats: null buffer");
    hipLaunchKernelGGL(episode_stats_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, synth_view(state, *cfg).fin, (int)cfg->num_envs, out4,
                       (int)reset);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_synth_last_infos(void *state, const pfa_synth_config *cfg, uint8_t *finished, double *episode_return,
                                    int32_t *episode_length, double *score, pfa_stream_t stream) {
    if (int rc = check_synth_any(cfg)) return rc;
    PFA_REQUIRE(state && finished && episode_return && episode_length && score, "synth.last_infos: null buffer");
    hipLaunchKernelGGL(episode_infos_kernel, dim3((unsigned)((cfg->num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       synth_view(state, *cfg).fin, (int)cfg->num_envs, finished, episode_return, (int *)episode_length, score);
    PFA_LAUNCH_CHECK();
    return 0;
}
