// lstm_fused.hip — the recurrent policy in rollout mode as fused MFMA kernels (lstm_tile.hpp):
//   lstm_pack_kernel             Wcat = [W_ih | W_hh] -> per-wave A-fragment order (512 KB, once per weight version)
//   lstm_policy_step_kernel      RecurrentPolicy.forward with action=None (frameworks/cleanrl.py:84-93 ->
//                                models.py:86-111): encode -> one nn.LSTM step -> decode -> sample_logits, state in/out
//   rollout_lstm_squared_kernel  clean_pufferl.evaluate (clean_pufferl.py:76-154) for a Squared vecenv and the
//                                recurrent policy: one persistent workgroup per 16 envs for all T steps; env state in
//                                registers, observation grid / [xe | h] tile in LDS, c in registers, the gate matrix
//                                streamed from L2 every step.  The LSTM state is carried across steps and across
//                                rollouts and is NOT reset on done, like the reference.
// Both kernels run the same device code per tile, so the protocol path and the fused rollout agree bit for bit.
// Roofline: fp32 MFMA — 2*(64*128 + 256*512 + 128*16) = 282,624 flop per env step (93% in the gate product).
#include "common.hpp"
#include "lane_ops.hpp"
#include "lstm_tile.hpp"
#include "mlp_tile.hpp"
#include "philox.hpp"
#include "sampler.hpp"
#include "squared_env.hpp"
#include "memory_env.hpp"
#include "synth_env.hpp"

namespace pfa {

// dst float4 index ((w*16 + kq)*8 + ct)*64 + lane  <-  Wcat[col(w, ct, c)][16kq + 4g .. +3]
__global__ void __launch_bounds__(256) lstm_pack_kernel(const float *__restrict__ params, int dp, int a, float4 *__restrict__ dst) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= kGatePackFloats / 4) return;
    dst[idx] = lstm_pack_fwd_elem(params, dp, a, idx);
}

// Shared LDS block of the two kernels.
template <int DP>
struct LstmLds {
    float xs[XTile<DP>::kFloats];   // observation tile (B operand of the encoder; the env writes its grids here)
    float xh[2][kXHTile];           // [xe | h] tiles; h_t lives in tile t&1, h_{t+1} is written into the other one
    float part[4][kPartFloats];       // head partials per wave
    float gbias[kLG];               // b_ih + b_hh
};

// One policy step on the tile: xs (obs) + xh[cur] (h in its h half) + cst -> h' into xh[cur^1], c' in cst, head partials
// in part.  Contains three workgroup barriers; the caller must barrier before reading part / rewriting xs.
// `after_prefetch()` runs right after the first weight groups are requested: global STORES of the caller go there, behind
// the loads in the in-order memory queue, so the weight ring never waits for a store acknowledgement.
template <int DP, class AfterPrefetch>
__device__ __forceinline__ void lstm_tile_step(const LstmFrags<DP> &w, const float4 *__restrict__ wp, LstmLds<DP> &L, int cur,
                                               f32x4 (&cst)[2], AfterPrefetch &&after_prefetch) {
    const int wv = wave_id(), c = lane_id() & 15, g = lane_id() >> 4;
    float4 ring[kGateRing][8];
    const Stream ws = stream_begin(wp);
    ring_prefetch<8, kGateRing>(ws, ring);
    after_prefetch();
    lstm_encode<DP>(w, L.xs, L.xh[cur]);
    __syncthreads();
    f32x4 acc[1][8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) acc[0][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    stream_product<16, 8, 1, kXHTile, kGateRing>(ws, L.xh[cur] + c * kXHS + 4 * g, ring, acc, NoSide{});
    f32x4 hn[2];
    lstm_cell(acc[0], L.gbias, cst, hn);
    float4 *dst = reinterpret_cast<float4 *>(L.xh[cur ^ 1] + c * kXHS + kLH + 32 * wv + 4 * g);
    dst[0] = make_float4(hn[0][0], hn[0][1], hn[0][2], hn[0][3]);
    dst[4] = make_float4(hn[1][0], hn[1][1], hn[1][2], hn[1][3]);
    __syncthreads();
    lstm_heads<DP>(w, L.xh[cur ^ 1], L.part);
    __syncthreads();
}

__device__ __forceinline__ LaneSample lstm_sample(const float (*part)[kPartFloats], int le, int lo, int a, float q) {
    const int i = lo * kPartStride + le;
    const float mine = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
    return sample_row16(mine, lo, a, q);
}
__device__ __forceinline__ LaneSample lstm_sample_heads(const float (*part)[kPartFloats], int le, int lo, int a, uint32_t heads, float q) {
    const int i = lo * kPartStride + le;
    const float mine = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
    return sample_row16_heads(mine, lo, a, heads, q);
}

// ---------------------------------------------------------------------------------------------
// policy(obs, (h, c)) in rollout mode over `rows` rows; h, c [rows][128] are updated in place.
// ---------------------------------------------------------------------------------------------
template <int DP>
__global__ void __launch_bounds__(kLstmThreads) lstm_policy_step_kernel(const float *obs, long long rows, const float *params,
                                                                       int a, uint32_t heads, const float4 *wpack, float *h,
                                                                       float *cell, const float *noise, uint64_t seed,
                                                                       uint64_t step, long long row_offset, long long *actions,
                                                                       float *logprob, float *entropy, float *value) {
    __shared__ LstmLds<DP> L;
    LstmFrags<DP> w;
    w.load(params, a);
    stage_gate_bias(params, DP, a, L.gbias);
    const float4 *wp = wpack + (size_t)__builtin_amdgcn_readfirstlane(wave_id()) * 16 * 8 * 64;  // wave-uniform: SGPR base + lane offset
    const int le = threadIdx.x >> 4, lo = threadIdx.x & 15;
    const int c = lane_id() & 15;
    const long long tiles = (rows + 15) / 16;
    for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        __syncthreads();
        lstm_stage_obs<DP>(obs, tile * 16, rows, L.xs);
        load_hstate(h, tile * 16, rows, L.xh[0]);
        f32x4 cst[2];
        load_cstate(cell, tile * 16 + c, tile * 16 + c < rows, cst);
        __syncthreads();
        lstm_tile_step<DP>(w, wp, L, 0, cst, [] {});
        store_cstate(cell, tile * 16 + c, tile * 16 + c < rows, cst);
        store_hstate(h, tile * 16, rows, L.xh[1]);
        const long long row = tile * 16 + le;
        const bool ok = row < rows;
        const float q = ok ? noise_lane(noise ? noise + row * a : nullptr, seed, step, (uint64_t)(row_offset + row), lo, a) : 1.0f;
        const LaneSample sm = heads ? lstm_sample_heads(L.part, le, lo, a, heads, q) : lstm_sample(L.part, le, lo, a, q);
        if (ok && lo == 0) {
            actions[row] = sm.action;
            logprob[row] = sm.logprob;
            if (entropy) entropy[row] = sm.entropy;
            value[row] = sm.value;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fused persistent rollout (see the header comment); structure follows rollout_mlp_squared_kernel (rollout.hip)
// ---------------------------------------------------------------------------------------------
template <int DP>
__global__ void __launch_bounds__(kLstmThreads) rollout_lstm_squared_kernel(SquaredView v, const float *params, int a,
                                                                           const float4 *wpack, float *h, float *cell,
                                                                           pfa_experience ex, const float *noise, uint64_t seed,
                                                                           uint64_t step0, long long env_offset, float *live_obs,
                                                                           float *live_rew, uint8_t *live_term,
                                                                           uint8_t *live_trunc, uint8_t *live_mask) {
    constexpr int XS = XTile<DP>::XS;
    __shared__ LstmLds<DP> L;
    __shared__ uint16_t tg[16 * kMaxTargets];
    const int le = threadIdx.x >> 4, lo = threadIdx.x & 15;  // sampling role: local env, output index
    const int e = blockIdx.x * 16 + le;
    const bool env_ok = e < v.n;
    const bool owner = lo == 0 && env_ok;
    const int T = ex.horizon_T;
    const int c = lane_id() & 15;
    const long long first = (long long)blockIdx.x * 16;

    LstmFrags<DP> w;
    w.load(params, a);
    stage_gate_bias(params, DP, a, L.gbias);
    const float4 *wp = wpack + (size_t)__builtin_amdgcn_readfirstlane(wave_id()) * 16 * 8 * 64;  // wave-uniform: SGPR base + lane offset

    lstm_stage_obs<DP>(live_obs, first, v.n, L.xs);
    load_hstate(h, first, v.n, L.xh[0]);
    f32x4 cst[2];
    load_cstate(cell, first + c, first + c < v.n, cst);
    SquaredEnv s;
    float reward = 0.0f;
    bool terminal = false;
    if (owner) {
        squared_load(v, e, s);
        for (int t = 0; t < v.nt; ++t) tg[le * kMaxTargets + t] = v.tgt[(size_t)t * v.n + e];
        reward = live_rew[e];
        terminal = live_term[e] != 0;
    }
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const int cur = t & 1;
        // Experience.store of the observation rows (clean_pufferl.py:443), env-major: row (e, t) at e*T + t
        lstm_tile_step<DP>(w, wp, L, cur, cst,
                           [&] { lstm_unstage_obs<DP>(L.xs, ex.obs + (size_t)t * DP, first, v.n, (size_t)T * DP); });
        const float q = env_ok ? noise_lane(noise ? noise + ((size_t)t * v.n + e) * a : nullptr, seed, step0 + t,
                                            (uint64_t)(env_offset + e), lo, a)
                               : 1.0f;
        const LaneSample sm = lstm_sample(L.part, le, lo, a, q);
        if (owner) {
            const size_t row = (size_t)e * T + t;
            ex.rewards[row] = reward;
            ex.dones[row] = terminal ? 1.0f : 0.0f;
            ex.actions[row] = sm.action;
            ex.logprobs[row] = sm.logprob;
            ex.values[row] = sm.value;
            float *grid = L.xs + le * XS;
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
        __syncthreads();
    }

    if (owner) {
        squared_store(v, e, s);
        for (int t = 0; t < v.nt; ++t) v.tgt[(size_t)t * v.n + e] = tg[le * kMaxTargets + t];
        v.fin[e] = 0;
        live_rew[e] = reward;
        live_term[e] = terminal ? 1 : 0;
        live_trunc[e] = 0;
        live_mask[e] = 1;
    }
    lstm_unstage_obs<DP>(L.xs, live_obs, first, v.n, (size_t)DP);
    store_cstate(cell, first + c, first + c < v.n, cst);
    store_hstate(h, first, v.n, L.xh[T & 1]);
}

// The same persistent rollout over the ocean Memory vecenv (memory_env.hpp; observation rows of 16 floats, one real column):
// the env that NEEDS the recurrent state.  Per step the owner thread of an env stores the scalars of the row, then either
// starts the next episode from the solution tape (auto-reset row) or applies memory_step to the sampled action and rewrites
// column 0 of the env's LDS observation row.
__global__ void __launch_bounds__(kLstmThreads) rollout_lstm_memory_kernel(MemoryView v, const float *params, int a,
                                                                          const float4 *wpack, float *h, float *cell,
                                                                          pfa_experience ex, const float *noise, uint64_t seed,
                                                                          uint64_t step0, long long env_offset, float *live_obs,
                                                                          float *live_rew, uint8_t *live_term,
                                                                          uint8_t *live_trunc, uint8_t *live_mask) {
    constexpr int DP = kMemDP, XS = XTile<DP>::XS;
    __shared__ LstmLds<DP> L;
    const int le = threadIdx.x >> 4, lo = threadIdx.x & 15;
    const int e = blockIdx.x * 16 + le;
    const bool env_ok = e < v.n;
    const bool owner = lo == 0 && env_ok;
    const int T = ex.horizon_T;
    const int c = lane_id() & 15;
    const long long first = (long long)blockIdx.x * 16;

    LstmFrags<DP> w;
    w.load(params, a);
    stage_gate_bias(params, DP, a, L.gbias);
    const float4 *wp = wpack + (size_t)__builtin_amdgcn_readfirstlane(wave_id()) * 16 * 8 * 64;

    lstm_stage_obs<DP>(live_obs, first, v.n, L.xs);
    load_hstate(h, first, v.n, L.xh[0]);
    f32x4 cst[2];
    load_cstate(cell, first + c, first + c < v.n, cst);
    MemoryEnv s = {};
    float reward = 0.0f;
    bool terminal = false;
    int last_fin = 0;
    if (owner) {
        s = v.env[e];
        reward = live_rew[e];
        terminal = live_term[e] != 0;
    }
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const int cur = t & 1;
        lstm_tile_step<DP>(w, wp, L, cur, cst,
                           [&] { lstm_unstage_obs<DP>(L.xs, ex.obs + (size_t)t * DP, first, v.n, (size_t)T * DP); });
        const float q = env_ok ? noise_lane(noise ? noise + ((size_t)t * v.n + e) * a : nullptr, seed, step0 + t,
                                            (uint64_t)(env_offset + e), lo, a)
                               : 1.0f;
        const LaneSample sm = lstm_sample(L.part, le, lo, a, q);
        if (owner) {
            const size_t row = (size_t)e * T + t;
            ex.rewards[row] = reward;
            ex.dones[row] = terminal ? 1.0f : 0.0f;
            ex.actions[row] = sm.action;
            ex.logprobs[row] = sm.logprob;
            ex.values[row] = sm.value;
            float o;
            last_fin = 0;
            if (s.done) {   // auto-reset row (vector.py:144-151): the action is ignored, the next solution comes off the tape
                if (s.rounds >= v.hdr->rounds_filled) v.hdr->underrun = 1;
                const uint32_t bits = v.tape[(size_t)(s.rounds % v.tape_rounds) * v.n + e];
                const long long rounds = s.rounds + 1;
                memory_begin_episode(s, bits, o, reward, terminal);
                s.rounds = rounds;
            } else {
                double fr, fs;
                int fl;
                if (memory_step(v, s, sm.action, o, reward, terminal, fr, fl, fs)) {
                    episode_account(v.fin[e], fr, fl, fs);
                    last_fin = 1;
                }
            }
            L.xs[le * XS] = o;
        }
        __syncthreads();
    }

    if (owner) {
        v.env[e] = s;
        v.fin[e].last_fin = last_fin;
        live_rew[e] = reward;
        live_term[e] = terminal ? 1 : 0;
        live_trunc[e] = 0;
        live_mask[e] = 1;
    }
    lstm_unstage_obs<DP>(L.xs, live_obs, first, v.n, (size_t)DP);
    store_cstate(cell, first + c, first + c < v.n, cst);
    store_hstate(h, first, v.n, L.xh[T & 1]);
}

// ... and over the synthetic byte-row env of BASELINE configs[2] (synth_env.hpp; rows of DP = obs_stride floats).  The 16 lanes
// of an env's sampling group regenerate its observation row after the step, 16 values per lane (one Philox call each).
template <int DP>
__global__ void __launch_bounds__(kLstmThreads) rollout_lstm_synth_kernel(SynthView v, const float *params, int a, const float4 *wpack,
                                                                         float *h, float *cell, pfa_experience ex, const float *noise,
                                                                         uint64_t seed, uint64_t step0, long long env_offset,
                                                                         float *live_obs, float *live_rew, uint8_t *live_term,
                                                                         uint8_t *live_trunc, uint8_t *live_mask) {
    constexpr int XS = XTile<DP>::XS;
    __shared__ LstmLds<DP> L;
    __shared__ int s_tick[16], s_episode[16];
    const int le = threadIdx.x >> 4, lo = threadIdx.x & 15;
    const int e = blockIdx.x * 16 + le;
    const bool env_ok = e < v.n;
    const bool owner = lo == 0 && env_ok;
    const int T = ex.horizon_T;
    const int c = lane_id() & 15;
    const long long first = (long long)blockIdx.x * 16;

    LstmFrags<DP> w;
    w.load(params, a);
    stage_gate_bias(params, DP, a, L.gbias);
    const float4 *wp = wpack + (size_t)__builtin_amdgcn_readfirstlane(wave_id()) * 16 * 8 * 64;

    lstm_stage_obs<DP>(live_obs, first, v.n, L.xs);
    load_hstate(h, first, v.n, L.xh[0]);
    f32x4 cst[2];
    load_cstate(cell, first + c, first + c < v.n, cst);
    SynthEnv s = {};
    float reward = 0.0f;
    bool terminal = false;
    int last_fin = 0;
    if (owner) {
        s = v.env[e];
        reward = live_rew[e];
        terminal = live_term[e] != 0;
    }
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const int cur = t & 1;
        lstm_tile_step<DP>(w, wp, L, cur, cst,
                           [&] { lstm_unstage_obs<DP>(L.xs, ex.obs + (size_t)t * DP, first, v.n, (size_t)T * DP); });
        const float q = env_ok ? noise_lane(noise ? noise + ((size_t)t * v.n + e) * a : nullptr, seed, step0 + t,
                                            (uint64_t)(env_offset + e), lo, a)
                               : 1.0f;
        const LaneSample sm = lstm_sample(L.part, le, lo, a, q);
        if (owner) {
            const size_t row = (size_t)e * T + t;
            ex.rewards[row] = reward;
            ex.dones[row] = terminal ? 1.0f : 0.0f;
            ex.actions[row] = sm.action;
            ex.logprobs[row] = sm.logprob;
            ex.values[row] = sm.value;
            last_fin = 0;
            if (s.done) {
                synth_begin_episode(s, reward, terminal);
            } else {
                double fr, fs;
                int fl;
                if (synth_step(v, s, sm.action, (int)L.xs[le * XS], reward, terminal, fr, fl, fs)) {
                    episode_account(v.fin[e], fr, fl, fs);
                    last_fin = 1;
                }
            }
            s_tick[le] = s.tick;
            s_episode[le] = s.episode;
        }
        __syncthreads();
        if (env_ok && lo * 16 < v.values) {   // the next observation row of this env, 16 values per lane
            float vals[16];
            synth_chunk(v, e, s_episode[le], s_tick[le], lo, vals);
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (lo * 16 + k < v.values) L.xs[le * XS + lo * 16 + k] = vals[k];
        }
        __syncthreads();
    }

    if (owner) {
        v.env[e] = s;
        v.fin[e].last_fin = last_fin;
        live_rew[e] = reward;
        live_term[e] = terminal ? 1 : 0;
        live_trunc[e] = 0;
        live_mask[e] = 1;
    }
    lstm_unstage_obs<DP>(L.xs, live_obs, first, v.n, (size_t)DP);
    store_cstate(cell, first + c, first + c < v.n, cst);
    store_hstate(h, first, v.n, L.xh[T & 1]);
}

int check_synth_config(const pfa_synth_config *c);   // synthetic.hip

static int check_lstm_dims(const pfa_mlp_dims *d) {
    PFA_REQUIRE(d != nullptr, "lstm: null dims");
    PFA_REQUIRE(d->hidden == kHidden, "lstm: hidden must be %d (got %d)", kHidden, d->hidden);
    PFA_REQUIRE(d->obs_stride == 16 || d->obs_stride == 32 || d->obs_stride == 64 || d->obs_stride == 96 || d->obs_stride == 128 ||
                    d->obs_stride == 160,
                "lstm: obs_stride must be one of 16/32/64/96/128/160 (got %d)", d->obs_stride);
    PFA_REQUIRE(d->num_actions >= 1 && d->num_actions <= 15, "lstm: num_actions must be in 1..15 (got %d)", d->num_actions);
    PFA_REQUIRE(d->heads == 0 || heads_count(d->heads, d->num_actions) >= 1, "lstm: head sizes 0x%x do not sum to num_actions %d",
                d->heads, d->num_actions);
    return 0;
}

#define PFA_LSTM_DISPATCH_DP(dp, CALL)            \
    switch (dp) {                                 \
        case 16: { constexpr int DP = 16; CALL; } break;   \
        case 32: { constexpr int DP = 32; CALL; } break;   \
        case 64: { constexpr int DP = 64; CALL; } break;   \
        case 96: { constexpr int DP = 96; CALL; } break;   \
        case 160: { constexpr int DP = 160; CALL; } break; \
        default: { constexpr int DP = 128; CALL; } break;  \
    }

}  // namespace pfa

using namespace pfa;

extern "C" int64_t pfa_lstm_param_count(const pfa_mlp_dims *dims) {
    if (check_lstm_dims(dims)) return -1;
    return lstm_offsets(dims->obs_stride, dims->num_actions).count;
}

extern "C" size_t pfa_lstm_pack_bytes(void) { return (size_t)kGatePackFloats * sizeof(float); }

extern "C" int pfa_lstm_pack(const float *params, const pfa_mlp_dims *dims, void *wpack, pfa_stream_t stream) {
    if (int rc = check_lstm_dims(dims)) return rc;
    PFA_REQUIRE(params && wpack, "lstm_pack: null buffer");
    PFA_REQUIRE((uintptr_t)wpack % 16 == 0, "lstm_pack: the packed buffer must be 16-byte aligned");
    hipLaunchKernelGGL(lstm_pack_kernel, dim3(kGatePackFloats / 4 / 256), dim3(256), 0, (hipStream_t)stream, params, dims->obs_stride,
                       dims->num_actions, (float4 *)wpack);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_lstm_policy_step(const float *obs, int64_t rows, const float *params, const pfa_mlp_dims *dims,
                                    const void *wpack, float *h, float *c, const float *noise, const pfa_noise_key *key,
                                    int64_t row_offset, int64_t *actions, float *logprob, float *entropy, float *value,
                                    pfa_stream_t stream) {
    if (int rc = check_lstm_dims(dims)) return rc;
    PFA_REQUIRE(rows >= 0, "lstm.step: negative rows");
    if (rows == 0) return 0;
    PFA_REQUIRE(obs && params && wpack && h && c && actions && logprob && value, "lstm.step: null buffer");
    PFA_REQUIRE(noise || key, "lstm.step: need an explicit noise tensor or a Philox key");
    const uint64_t seed = key ? key->seed : 0, step = key ? key->step : 0;
    const int64_t tiles = (rows + 15) / 16;
    const unsigned grid = (unsigned)(tiles < 2048 ? tiles : 2048);
    ScopedKernelTimer timer("lstm_policy_step", (hipStream_t)stream);
    PFA_LSTM_DISPATCH_DP(dims->obs_stride,
                         hipLaunchKernelGGL(lstm_policy_step_kernel<DP>, dim3(grid), dim3(kLstmThreads), 0, (hipStream_t)stream, obs,
                                            (long long)rows, params, dims->num_actions, dims->heads, (const float4 *)wpack, h, c,
                                            noise, seed, step, (long long)row_offset, (long long *)actions, logprob, entropy, value));
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_rollout_lstm_squared(void *state, const pfa_squared_config *cfg, const float *params,
                                        const pfa_mlp_dims *dims, const void *wpack, float *h, float *c,
                                        const pfa_experience *exp, const float *noise, const pfa_noise_key *key,
                                        int64_t env_offset, float *obs, float *rewards, uint8_t *terminals,
                                        uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    if (int rc = check_lstm_dims(dims)) return rc;
    PFA_REQUIRE(state && cfg && params && wpack && h && c && exp && obs && rewards && terminals && truncations && masks,
                "rollout_lstm: null buffer");
    PFA_REQUIRE(dims->heads == 0, "rollout_lstm: the fused rollout samples one Discrete head");
    PFA_REQUIRE(cfg->obs_stride == dims->obs_stride, "rollout_lstm: env obs_stride %d != policy obs_stride %d", cfg->obs_stride,
                dims->obs_stride);
    PFA_REQUIRE(exp->horizon_T >= 1, "rollout_lstm: horizon must be >= 1");
    PFA_REQUIRE(exp->obs && exp->actions && exp->logprobs && exp->values && exp->rewards && exp->dones,
                "rollout_lstm: null experience buffer");
    PFA_REQUIRE(noise || key, "rollout_lstm: need an explicit noise tensor or a Philox key");
    PFA_REQUIRE(cfg->num_targets <= kMaxTargets, "rollout_lstm: too many targets");
    SquaredView v = squared_view(state, *cfg);
    const uint64_t seed = key ? key->seed : 0, step = key ? key->step : 0;
    const unsigned grid = (unsigned)((cfg->num_envs + 15) / 16);
    ScopedKernelTimer timer("rollout_lstm_squared", (hipStream_t)stream);
    PFA_LSTM_DISPATCH_DP(dims->obs_stride,
                         hipLaunchKernelGGL(rollout_lstm_squared_kernel<DP>, dim3(grid), dim3(kLstmThreads), 0, (hipStream_t)stream,
                                            v, params, dims->num_actions, (const float4 *)wpack, h, c, *exp, noise, seed, step,
                                            (long long)env_offset, obs, rewards, terminals, truncations, masks));
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_rollout_lstm_memory(void *state, const pfa_memory_config *cfg, const float *params, const pfa_mlp_dims *dims,
                                       const void *wpack, float *h, float *c, const pfa_experience *exp, const float *noise,
                                       const pfa_noise_key *key, int64_t env_offset, float *obs, float *rewards, uint8_t *terminals,
                                       uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    if (int rc = check_lstm_dims(dims)) return rc;
    PFA_REQUIRE(state && cfg && params && wpack && h && c && exp && obs && rewards && terminals && truncations && masks,
                "rollout_lstm_memory: null buffer");
    PFA_REQUIRE(dims->heads == 0 && dims->num_actions == 2, "rollout_lstm_memory: ocean.Memory takes one Discrete(2) action");
    PFA_REQUIRE(dims->obs_stride == kMemDP, "rollout_lstm_memory: observation rows are %d floats (got %d)", kMemDP, dims->obs_stride);
    PFA_REQUIRE(cfg->num_envs >= 1 && cfg->mem_length >= 1 && cfg->mem_length <= kMemMaxLen && cfg->tape_rounds >= 2,
                "rollout_lstm_memory: bad env configuration");
    PFA_REQUIRE(exp->horizon_T >= 1 && exp->obs && exp->actions && exp->logprobs && exp->values && exp->rewards && exp->dones,
                "rollout_lstm_memory: bad experience buffers");
    PFA_REQUIRE(noise || key, "rollout_lstm_memory: need an explicit noise tensor or a Philox key");
    const uint64_t seed = key ? key->seed : 0, step = key ? key->step : 0;
    ScopedKernelTimer timer("rollout_lstm_memory", (hipStream_t)stream);
    hipLaunchKernelGGL(rollout_lstm_memory_kernel, dim3((unsigned)((cfg->num_envs + 15) / 16)), dim3(kLstmThreads), 0, (hipStream_t)stream,
                       memory_view(state, *cfg), params, dims->num_actions, (const float4 *)wpack, h, c, *exp, noise, seed, step,
                       (long long)env_offset, obs, rewards, terminals, truncations, masks);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_rollout_lstm_synth(void *state, const pfa_synth_config *cfg, const float *params, const pfa_mlp_dims *dims,
                                      const void *wpack, float *h, float *c, const pfa_experience *exp, const float *noise,
                                      const pfa_noise_key *key, int64_t env_offset, float *obs, float *rewards, uint8_t *terminals,
                                      uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    if (int rc = check_lstm_dims(dims)) return rc;
    if (int rc = check_synth_config(cfg)) return rc;
    PFA_REQUIRE(state && params && wpack && h && c && exp && obs && rewards && terminals && truncations && masks,
                "rollout_lstm_synth: null buffer");
    PFA_REQUIRE(dims->heads == 0 && dims->num_actions == cfg->num_actions, "rollout_lstm_synth: the policy must have one Discrete(%d) head",
                cfg->num_actions);
    PFA_REQUIRE(dims->obs_stride == cfg->obs_stride, "rollout_lstm_synth: env obs_stride %d != policy obs_stride %d", cfg->obs_stride,
                dims->obs_stride);
    PFA_REQUIRE(exp->horizon_T >= 1 && exp->obs && exp->actions && exp->logprobs && exp->values && exp->rewards && exp->dones,
                "rollout_lstm_synth: bad experience buffers");
    PFA_REQUIRE(noise || key, "rollout_lstm_synth: need an explicit noise tensor or a Philox key");
    const uint64_t seed = key ? key->seed : 0, step = key ? key->step : 0;
    const unsigned grid = (unsigned)((cfg->num_envs + 15) / 16);
    ScopedKernelTimer timer("rollout_lstm_synth", (hipStream_t)stream);
    PFA_LSTM_DISPATCH_DP(dims->obs_stride,
                         hipLaunchKernelGGL(rollout_lstm_synth_kernel<DP>, dim3(grid), dim3(kLstmThreads), 0, (hipStream_t)stream,
                                            synth_view(state, *cfg), params, dims->num_actions, (const float4 *)wpack, h, c, *exp, noise,
                                            seed, step, (long long)env_offset, obs, rewards, terminals, truncations, masks));
    PFA_LAUNCH_CHECK();
    return 0;
}
