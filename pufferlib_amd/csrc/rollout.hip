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
#include "mlp_tile.hpp"
#include "philox.hpp"
#include "squared_env.hpp"

namespace pfa {

// Exp(1) noise for (row, step): explicit tensor if given, else the Philox stream (philox.hpp).
__device__ __forceinline__ void noise_row(const float *noise_row_ptr, uint64_t seed, uint64_t step, uint64_t row, int a,
                                          float (&q)[15]) {
#pragma unroll
    for (int o = 0; o < 15; ++o) q[o] = 1.0f;
    if (noise_row_ptr) {
#pragma unroll
        for (int o = 0; o < 15; ++o)
            if (o < a) q[o] = noise_row_ptr[o];
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (4 * j < a) {
            const u32x4 w = philox4x32_10((uint32_t)row, (uint32_t)j, (uint32_t)step,
                                          (uint32_t)(step >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
            const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * j + i < 15) q[4 * j + i] = -logf(philox_uniform(ws[i]));
        }
    }
}

// Stage 16 rows x DP floats (global, row stride DP) into the padded LDS tile.
template <int DP>
__device__ __forceinline__ void stage_rows(const float *src, long long first_row, long long rows, float *xs) {
    constexpr int XS = XTile<DP>::XS;
    constexpr int V = DP / 4;  // float4 per row
    const int lane = lane_id();
#pragma unroll
    for (int j = 0; j < (16 * V + 63) / 64; ++j) {
        const int idx = lane + 64 * j;
        if (idx < 16 * V) {
            const int r = idx / V, c4 = idx - r * V;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (first_row + r < rows) v = *reinterpret_cast<const float4 *>(src + (first_row + r) * DP + 4 * c4);
            float2 *d = reinterpret_cast<float2 *>(xs + r * XS + 4 * c4);
            d[0] = make_float2(v.x, v.y);
            d[1] = make_float2(v.z, v.w);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// standalone forward + sample over `rows` observation rows
// ---------------------------------------------------------------------------------------------
template <int DP>
__global__ void __launch_bounds__(64) mlp_forward_sample_kernel(const float *obs, long long rows, const float *params, int a,
                                                               const float *noise, uint64_t seed, uint64_t step,
                                                               long long row_offset, long long *actions, float *logprob,
                                                               float *entropy, float *value) {
    __shared__ float xs[XTile<DP>::kFloats];
    MlpFwdFrags<DP> w;
    w.load(params, a);
    const long long tiles = (rows + 15) / 16;
    for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        __syncthreads();
        stage_rows<DP>(obs, tile * 16, rows, xs);
        __syncthreads();
        f32x4 h[kMT], out;
        mlp_forward_tile<DP>(w, xs, h, out);
        float logits[15], val;
        gather_row_outputs(out, a, logits, val);
        const long long row = tile * 16 + lane_id();
        if (lane_id() < 16 && row < rows) {
            float q[15];
            noise_row(noise ? noise + row * a : nullptr, seed, step, (uint64_t)(row_offset + row), a, q);
            const SampleOut s = sample_logits_row(logits, a, q);
            actions[row] = s.action;
            logprob[row] = s.logprob;
            if (entropy) entropy[row] = s.entropy;
            value[row] = val;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fused persistent rollout
// ---------------------------------------------------------------------------------------------
template <int DP>
__global__ void __launch_bounds__(64) rollout_mlp_squared_kernel(SquaredView v, const float *params, int a, pfa_experience ex,
                                                                const float *noise, uint64_t seed, uint64_t step0,
                                                                long long env_offset, float *live_obs, float *live_rew,
                                                                uint8_t *live_term, uint8_t *live_trunc,
                                                                uint8_t *live_mask) {
    constexpr int XS = XTile<DP>::XS;
    __shared__ float xs[XTile<DP>::kFloats];
    __shared__ uint16_t tg[16 * kMaxTargets];
    const int lane = lane_id();
    const int e = blockIdx.x * 16 + lane;  // env owned by lanes 0..15
    const bool owner = lane < 16 && e < v.n;
    const int T = ex.horizon_T;

    MlpFwdFrags<DP> w;
    w.load(params, a);

    // recv(): the live buffers are the current observation / reward / terminal of every env
    stage_rows<DP>(live_obs, (long long)blockIdx.x * 16, v.n, xs);
    SquaredEnv s;
    float reward = 0.0f;
    bool terminal = false;
    if (owner) {
        squared_load(v, e, s);
        for (int t = 0; t < v.nt; ++t) tg[lane * kMaxTargets + t] = v.tgt[(size_t)t * v.n + e];
        reward = live_rew[e];
        terminal = live_term[e] != 0;
    }
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        // Experience.store of the observation row (clean_pufferl.py:443), env-major
        {
            constexpr int V = DP / 4;
#pragma unroll
            for (int j = 0; j < (16 * V + 63) / 64; ++j) {
                const int idx = lane + 64 * j;
                if (idx < 16 * V) {
                    const int r = idx / V, c4 = idx - r * V;
                    const int er = blockIdx.x * 16 + r;
                    if (er < v.n) {
                        const float2 *sp = reinterpret_cast<const float2 *>(xs + r * XS + 4 * c4);
                        const float2 lo = sp[0], hi = sp[1];
                        *reinterpret_cast<float4 *>(ex.obs + ((size_t)er * T + t) * DP + 4 * c4) =
                            make_float4(lo.x, lo.y, hi.x, hi.y);
                    }
                }
            }
        }
        f32x4 h[kMT], out;
        mlp_forward_tile<DP>(w, xs, h, out);
        float logits[15], val;
        gather_row_outputs(out, a, logits, val);
        __syncthreads();  // all lanes are done reading xs before the env step rewrites it
        if (owner) {
            const size_t row = (size_t)e * T + t;
            float q[15];
            noise_row(noise ? noise + ((size_t)t * v.n + e) * a : nullptr, seed, step0 + t, (uint64_t)(env_offset + e), a, q);
            const SampleOut so = sample_logits_row(logits, a, q);
            ex.rewards[row] = reward;
            ex.dones[row] = terminal ? 1.0f : 0.0f;
            ex.actions[row] = so.action;
            ex.logprobs[row] = so.logprob;
            ex.values[row] = val;
            // send(): vector.py:144-151
            float *grid = xs + lane * XS;
            uint16_t *tc = tg + lane * kMaxTargets;
            if (s.done) {
                if ((long long)s.rounds >= v.hdr->rounds_filled) v.hdr->underrun = 1;
                const uint16_t *tr = v.tape + (size_t)(s.rounds % (uint32_t)v.tape_rounds) * v.nt * v.n;
                squared_reset(v, e, s, grid, tr, tc, reward, terminal);
                s.rounds += 1;
            } else {
                bool fin;
                double fr, fs;
                int fl;
                squared_step(v, s, grid, tc, so.action, reward, terminal, fin, fr, fl, fs);
            }
        }
        __syncthreads();
    }

    // write back: env state + the live buffers the next recv() returns
    if (owner) {
        squared_store(v, e, s);
        for (int t = 0; t < v.nt; ++t) v.tgt[(size_t)t * v.n + e] = tg[lane * kMaxTargets + t];
        v.fin[e] = 0;
        live_rew[e] = reward;
        live_term[e] = terminal ? 1 : 0;
        live_trunc[e] = 0;
        live_mask[e] = 1;
    }
    {
        constexpr int V = DP / 4;
#pragma unroll
        for (int j = 0; j < (16 * V + 63) / 64; ++j) {
            const int idx = lane + 64 * j;
            if (idx < 16 * V) {
                const int r = idx / V, c4 = idx - r * V;
                const int er = blockIdx.x * 16 + r;
                if (er < v.n) {
                    const float2 *sp = reinterpret_cast<const float2 *>(xs + r * XS + 4 * c4);
                    const float2 lo = sp[0], hi = sp[1];
                    *reinterpret_cast<float4 *>(live_obs + (size_t)er * DP + 4 * c4) = make_float4(lo.x, lo.y, hi.x, hi.y);
                }
            }
        }
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
    return 0;
}

#define PFA_DISPATCH_DP(dp, CALL)                 \
    switch (dp) {                                 \
        case 16: { constexpr int DP = 16; CALL; } break;   \
        case 32: { constexpr int DP = 32; CALL; } break;   \
        case 64: { constexpr int DP = 64; CALL; } break;   \
        case 96: { constexpr int DP = 96; CALL; } break;   \
        default: { constexpr int DP = 128; CALL; } break;  \
    }

}  // namespace pfa

using namespace pfa;

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
    PFA_DISPATCH_DP(dims->obs_stride,
                    hipLaunchKernelGGL(mlp_forward_sample_kernel<DP>, dim3(grid), dim3(64), 0, (hipStream_t)stream, obs,
                                       (long long)rows, params, dims->num_actions, noise, seed, step, (long long)row_offset,
                                       (long long *)actions, logprob, entropy, value));
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_rollout_mlp_squared(void *state, const pfa_squared_config *cfg, const float *params,
                                       const pfa_mlp_dims *dims, const pfa_experience *exp, const float *noise,
                                       const pfa_noise_key *key, int64_t env_offset, float *obs, float *rewards,
                                       uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    if (int rc = check_dims(dims)) return rc;
    PFA_REQUIRE(state && cfg && params && exp && obs && rewards && terminals && truncations && masks, "rollout: null buffer");
    PFA_REQUIRE(cfg->obs_stride == dims->obs_stride, "rollout: env obs_stride %d != policy obs_stride %d", cfg->obs_stride,
                dims->obs_stride);
    PFA_REQUIRE(exp->horizon_T >= 1, "rollout: horizon must be >= 1");
    PFA_REQUIRE(exp->obs && exp->actions && exp->logprobs && exp->values && exp->rewards && exp->dones,
                "rollout: null experience buffer");
    PFA_REQUIRE(noise || key, "rollout: need an explicit noise tensor or a Philox key");
    PFA_REQUIRE(cfg->num_targets <= kMaxTargets, "rollout: too many targets");
    SquaredView v = squared_view(state, *cfg);
    const uint64_t seed = key ? key->seed : 0, step = key ? key->step : 0;
    const unsigned grid = (unsigned)((cfg->num_envs + 15) / 16);
    ScopedKernelTimer timer("rollout_mlp_squared", (hipStream_t)stream);
    PFA_DISPATCH_DP(dims->obs_stride,
                    hipLaunchKernelGGL(rollout_mlp_squared_kernel<DP>, dim3(grid), dim3(64), 0, (hipStream_t)stream, v, params,
                                       dims->num_actions, *exp, noise, seed, step, (long long)env_offset, obs, rewards,
                                       terminals, truncations, masks));
    PFA_LAUNCH_CHECK();
    return 0;
}
