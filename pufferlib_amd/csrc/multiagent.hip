// multiagent.hip — ocean `Multiagent` (pufferlib/environments/ocean/ocean.py:148-224) as a device-resident vecenv (SURVEY.md
// §8f rank 2): the two-agents-per-env case of the backend protocol.  Reference stack per env: pufferlib.vector.Serial
// (vector.py:78-162) over make_multiagent (ocean/environment.py:76-79) = PettingZooPufferEnv (emulation.py:236-420) +
// MultiagentEpisodeStats + ocean.Multiagent.
//
// Agent rows are env-major: row 2e is agent 1 of env e (observation 0, rewarded for action 0), row 2e+1 is agent 2
// (observation 1, rewarded for action 1).  Every step terminates both agents (`terminal = {1: True, 2: True}`), so the next
// send is the reset row of the whole env (emulation.py:283-284 `done` = all agents done; vector.py:147-149).  The infos the
// reference returns are the env's own `{agent: {'score': reward}}` (MultiagentEpisodeStats builds its episode dict and drops
// it, postprocess.py:159-177), i.e. per agent slot the mean score — accumulated here as four exact integer-valued sums.
#include "common.hpp"

namespace pfa {

constexpr int kMultiDP = 16;  // observation row stride in floats (1 real column)

struct MultiagentView {
    double *sums;  // [4]: finished steps of agent 1, its score sum, the same for agent 2
    int *done;     // [num_envs]
    int n;         // envs
};
__host__ __device__ inline size_t multiagent_state_bytes(int n) { return 32 + (size_t)n * sizeof(int); }
__host__ __device__ inline MultiagentView multiagent_view(void *state, int n) {
    MultiagentView v;
    v.sums = (double *)state;
    v.done = (int *)((char *)state + 32);
    v.n = n;
    return v;
}

__global__ void __launch_bounds__(256) multiagent_reset_kernel(MultiagentView v, float *obs, float *rewards, uint8_t *terminals,
                                                              uint8_t *truncations, uint8_t *masks) {
    const int a = blockIdx.x * 256 + threadIdx.x;  // agent row
    if (a < 4) v.sums[a] = 0.0;
    if (a >= 2 * v.n) return;
    if ((a & 1) == 0) v.done[a >> 1] = 0;
#pragma unroll
    for (int k = 1; k < kMultiDP; ++k) obs[(size_t)a * kMultiDP + k] = 0.0f;
    obs[(size_t)a * kMultiDP] = (float)(a & 1);
    rewards[a] = 0.0f;
    terminals[a] = 0;
    truncations[a] = 0;
    masks[a] = 1;
}

// One thread per agent row; both rows of an env read the env's done flag before either writes it (the even row writes it
// after the wave-level exchange below: rows 2e and 2e+1 are adjacent lanes of one wavefront).
__global__ void __launch_bounds__(256) multiagent_send_kernel(MultiagentView v, const long long *actions, float *obs, float *rewards,
                                                             uint8_t *terminals, uint8_t *truncations, uint8_t *masks) {
    const int a = blockIdx.x * 256 + threadIdx.x;
    const bool live = a < 2 * v.n;
    const int e = a >> 1, slot = a & 1;
    int stepped = 0, score = 0;
    if (live) {
        const int was_done = v.done[e];
        if (!was_done) {  // ocean.py:187-205: agent 1 scores with action 0, agent 2 with action 1
            stepped = 1;
            score = (int)actions[a] == slot;
        }
        obs[(size_t)a * kMultiDP] = (float)slot;
        rewards[a] = (float)score;
        terminals[a] = (uint8_t)stepped;
        truncations[a] = 0;
        masks[a] = 1;
    }
    __syncthreads();  // every read of done[] in this workgroup (both rows of an env share it) precedes the writes
    if (live && slot == 0) v.done[e] = stepped;
    // per-slot sums: wave reduction, then one atomic per wave and slot (integer-valued doubles: order-independent, exact)
    unsigned long long m_step = __ballot(stepped), m_score = __ballot(score);
    const unsigned long long even = 0x5555555555555555ull;
    if ((threadIdx.x & 63) == 0 && m_step) {
        atomicAdd(&v.sums[0], (double)__popcll(m_step & even));
        atomicAdd(&v.sums[1], (double)__popcll(m_score & even));
        atomicAdd(&v.sums[2], (double)__popcll(m_step & ~even));
        atomicAdd(&v.sums[3], (double)__popcll(m_score & ~even));
    }
}

__global__ void multiagent_stats_kernel(MultiagentView v, double *out4, int reset) {
    const int k = threadIdx.x;
    if (k < 4) {
        out4[k] = v.sums[k];
        if (reset) v.sums[k] = 0.0;
    }
}

}  // namespace pfa

using namespace pfa;

extern "C" size_t pfa_multiagent_state_bytes(int32_t num_envs) { return num_envs > 0 ? multiagent_state_bytes(num_envs) : 0; }

extern "C" int pfa_multiagent_async_reset(void *state, int32_t num_envs, float *obs, float *rewards, uint8_t *terminals,
                                          uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    PFA_REQUIRE(state && num_envs >= 1 && obs && rewards && terminals && truncations && masks, "multiagent.async_reset: bad arguments");
    const int rows = 2 * num_envs;
    hipLaunchKernelGGL(multiagent_reset_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       multiagent_view(state, num_envs), obs, rewards, terminals, truncations, masks);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_multiagent_send(void *state, int32_t num_envs, const int64_t *actions, float *obs, float *rewards,
                                   uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    PFA_REQUIRE(state && num_envs >= 1 && actions && obs && rewards && terminals && truncations && masks,
                "multiagent.send: bad arguments");
    const int rows = 2 * num_envs;
    hipLaunchKernelGGL(multiagent_send_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       multiagent_view(state, num_envs), (const long long *)actions, obs, rewards, terminals, truncations, masks);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_multiagent_episode_stats(void *state, int32_t num_envs, double *out4, int32_t reset, pfa_stream_t stream) {
    PFA_REQUIRE(state && num_envs >= 1 && out4, "multiagent.episode_stats: bad arguments");
    hipLaunchKernelGGL(multiagent_stats_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, multiagent_view(state, num_envs), out4,
                       (int)reset);
    PFA_LAUNCH_CHECK();
    return 0;
}
