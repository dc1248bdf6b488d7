/* pufferlib_amd.h — C ABI of the MI355X-native PPO rollout-and-update engine.
 *
 * The reference (PufferLib 1.0.1) has exactly one native entry point on this path,
 * c_gae.compute_gae (c_gae.pyx:11-32); everything else is Python calling stock PyTorch ops.
 * This header freezes the C ABI a maintainer would bind (ctypes / cffi / Cython) to replace the
 * body of each reference function named below.  Conventions:
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - nothing here allocates or frees caller memory; sizes come from the *_bytes() helpers;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); calls only enqueue work;
 *   - return 0 on success, negative on error (pfa_last_error() gives the message);
 *   - no torch types; PyTorch is only the allocator / stream owner on the Python side.
 * gfx950 only.  All arithmetic fp32 unless stated (fp64 for episode-return sums, python-float parity).
 */
#ifndef PUFFERLIB_AMD_H
#define PUFFERLIB_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *pfa_stream_t;

int pfa_version(void);
const char *pfa_last_error(void);

/* Optional measurement hook (bench.py roofline leg).  mode 0 = off, 1 = only the dominant kernel
 * ("ppo_mlp_grad"), 2 = every instrumented kernel.  When enabled, the launches of the named kernels
 * ("rollout_mlp_squared", "gae", "ppo_mlp_grad", "ppo_reduce", "adam_clip", "squared_tape") are bracketed by
 * hipEvents recorded on the launch stream.  pfa_timing_read synchronises on the recorded events and returns
 * the launch count and the summed device time in ms (HOST pointers). */
int pfa_timing_enable(int mode);
int pfa_timing_select(const char *kernel); /* the kernel mode 1 times (default "ppo_mlp_grad") */
int pfa_timing_stride(int every);          /* mode 1 brackets every `every`-th launch of that kernel (default 1 = all) */
int pfa_timing_reset(void);
int pfa_timing_read(const char *kernel, int64_t *launches_host, double *total_ms_host);

/* ------------------------------------------------------------------------------------------
 * GAE — replaces c_gae.compute_gae (c_gae.pyx:11-32), called from clean_pufferl.py:168-169.
 * One reverse affine scan over the whole flat (env-major) batch, crossing env boundaries exactly
 * like the reference; advantages[n-1] = 0.  Contract: within 1e-5 of the reference's sequential fp32 loop, always.  In practice much
 * closer: every thread warms up on the elements behind its own (csrc/gae.hip gae_exact_kernel) so that it sits ON the reference's
 * rounded sequence, and for gamma * gae_lambda up to ~0.985 the result is the reference's bit pattern on typical data (one launch up to
 * 0.984: the walkers start from zero inside a 1024- or 2048-element window; two launches with f64 chunk maps above) (every case the
 * tests hold, 524 288 rows, done rates 0 - 25 %); the argument is a contraction in real arithmetic, so isolated entries may sit
 * 1 ulp off on data with an extreme dynamic range inside one 1024-element window, and a few ulps off above ~0.985.
 * `returns` (nullable) = advantages + values
 * (clean_pufferl.py:482).  workspace >= pfa_gae_workspace_bytes(n).
 * ------------------------------------------------------------------------------------------ */
size_t pfa_gae_workspace_bytes(int64_t n);
int pfa_gae_f32(const float *dones, const float *values, const float *rewards, float *advantages,
                float *returns, int64_t n, float gamma, float gae_lambda, void *workspace,
                pfa_stream_t stream);
/* compute_gae together with what the update and its log line need from the advantages, in one pass over the rows
 * (replaces pfa_gae_f32 + pfa_ppo_adv_stats + the sums of pfa_train_log_sums: three launches instead of six):
 *   adv_stats[2 m + k]  sum adv, sum adv^2 of minibatch m's rows (clean_pufferl.py:211-213 normalisation; minibatch m =
 *                       segments {m + k nmb} of bptt_horizon rows, clean_pufferl.py:455-457), f64
 *   ev4[0..3]           sum y_true, sum y_true^2, sum adv, sum adv^2 with y_true = adv + values in storage (step-major)
 *                       order — the reference's explained-variance inputs (clean_pufferl.py:266-270, 476), f64
 *   zero8 (nullable)    eight doubles cleared by the last launch (the caller's running loss sums)
 * Fixed-order f64 reductions (deterministic).  Shapes: bptt_horizon a multiple of 8, (bptt_horizon / 8) * num_minibatches a
 * power of two <= 128, num_minibatches <= 32, n a multiple of num_envs and of num_minibatches * bptt_horizon —
 * pfa_gae_sums_supported says; other shapes use the separate entry points.  workspace >= pfa_gae_sums_workspace_bytes. */
int pfa_gae_sums_supported(int64_t n, int32_t num_envs, int32_t num_minibatches, int32_t bptt_horizon);
size_t pfa_gae_sums_workspace_bytes(int64_t n, int32_t num_minibatches);
int pfa_gae_sums_f32(const float *dones, const float *values, const float *rewards, float *advantages, float *returns,
                     int64_t n, float gamma, float gae_lambda, int32_t num_envs, int32_t num_minibatches, int32_t bptt_horizon,
                     double *adv_stats, double *ev4, double *zero8, void *workspace, pfa_stream_t stream);
/* Data-parallel forms: the array is one rank's SHARD of the reference's single flat batch (rank-major order; c_gae.pyx:11-32 runs
 * across env boundaries, so it also runs across shard boundaries).
 *
 * Halo form (the one clean_pufferl runs; csrc/gae.hip gae_halo_*): every rank publishes the bit patterns of its first
 * min(n, H) rows, H = pfa_gae_halo_rows(gamma, gae_lambda), next to `n_extra` other f64 sums of the caller (episode statistics);
 * ONE all-reduce(SUM) of out[n_extra + 3 world min(n, H)]; every rank drops the rows that follow its shard behind its arrays
 * (which hold n + H elements) and runs the single-rank kernel over them: advantages are the BITS of the single flat scan.
 *   pfa_gae_halo_rows      H (a multiple of 8, <= 2056); 0 = gamma * gae_lambda > 0.984: use the f64-carry form below
 *   pfa_gae_halo_publish   out[0 .. n_extra) = extra, then [world][3][min(n, H)] bit patterns as f64 (zeros in the peers' places)
 *   pfa_gae_halo_unpack    gathered = out + n_extra; returns halo_len = min(H, rows that follow this shard) (>= 0), < 0 on error
 *   pfa_gae_halo_f32       the scan; adv_stats != NULL: + this rank's share of pfa_gae_sums_f32's sums (same shape rules) */
int32_t pfa_gae_halo_rows(float gamma, float gae_lambda);
int pfa_gae_halo_publish(const float *dones, const float *values, const float *rewards, int64_t n, float gamma, float gae_lambda,
                         const double *extra, int32_t n_extra, double *out, int32_t rank, int32_t world, pfa_stream_t stream);
int pfa_gae_halo_unpack(const double *gathered, int32_t rank, int32_t world, int64_t n, float gamma, float gae_lambda, float *dones,
                        float *values, float *rewards, pfa_stream_t stream);
int pfa_gae_halo_f32(const float *dones, const float *values, const float *rewards, float *advantages, float *returns, int64_t n,
                     int32_t halo_len, float gamma, float gae_lambda, int32_t num_envs, int32_t num_minibatches, int32_t bptt_horizon,
                     double *adv_stats, double *ev4, double *zero8, void *workspace, pfa_stream_t stream);
/* f64-carry form (any gamma * gae_lambda; within a few fp32 ulps of the flat scan): pass 2 finishes the scan from the block
 * aggregates in `workspace` and carry_in[1] (f64, device) = the advantage of the first element after this shard.
 * has_next != 0: a later shard exists, the three input arrays hold n+1 readable elements (element n = the next shard's first
 * row) and the last element is an interior row; has_next == 0: last shard, advantages[n-1] = 0 and carry_in is ignored. */
int pfa_gae_shard_pass2(const float *dones, const float *values, const float *rewards, float *advantages,
                        float *returns, int64_t n, int has_next, float gamma, float gae_lambda,
                        const void *workspace, const double *carry_in, pfa_stream_t stream);
/* ... with ONE exchange (csrc/gae.hip): a shard's affine map is (interior) o (last element) and only the last
 * element needs the next shard's first row, so every rank publishes six numbers it can compute from its own rows — interior map
 * (C, D), values[n-1], first row (done, value, reward) — next to `n_extra` other f64 sums of the caller (episode statistics):
 *   pfa_gae_shard_publish   out[0 .. n_extra) = extra, out[n_extra + 6 q + j] = the six numbers for q == rank, 0 elsewhere
 *   (caller) ONE all-reduce(SUM) of out[n_extra + 6 world]
 *   pfa_gae_shard_fold      gathered = out + n_extra: completes every later shard's map, folds them into carry_out[0], patches this
 *                           shard's last block aggregate in `workspace`, writes the halo row at index n of dones / values / rewards
 *   pfa_gae_shard_pass2     has_next = rank < world - 1, carry_in = carry_out.
 * `workspace` (pfa_gae_workspace_bytes(n)) must not be touched between the three calls; the arrays hold n + 1 elements. */
int pfa_gae_shard_publish(const float *dones, const float *values, const float *rewards, int64_t n, float gamma, float gae_lambda,
                          void *workspace, const double *extra, int32_t n_extra, double *out, int32_t rank, int32_t world,
                          pfa_stream_t stream);
int pfa_gae_shard_fold(const double *gathered, int32_t rank, int32_t world, int64_t n, float gamma, float gae_lambda, void *workspace,
                       float *dones, float *values, float *rewards, double *carry_out, pfa_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * Squared vecenv — replaces pufferlib.vector.Serial (vector.py:70-166) over
 * ocean.environment.make_squared (ocean/environment.py:28-31), i.e. GymnasiumPufferEnv
 * (emulation.py:169-228) + EpisodeStats (postprocess.py:18-54) + ocean.Squared (ocean.py:406-513),
 * including the process-global MT19937 `random.sample` stream shared by all envs in index order.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t num_envs;           /* N */
    int32_t distance_to_target; /* d; grid is (2d+1)^2 */
    int32_t num_targets;        /* nt >= 1 (already resolved from -1 -> 4d) */
    int32_t obs_stride;         /* floats per observation row in every obs buffer (>= (2d+1)^2) */
    int32_t tape_rounds;        /* capacity of the reset-target tape, in reset rounds */
} pfa_squared_config;

size_t pfa_squared_state_bytes(const pfa_squared_config *cfg);
/* Serial.async_reset(seed) (vector.py:112-135): env i reseeds the shared generator with seed+i and
 * resets.  Writes the live buffers recv() returns: obs [N][obs_stride] f32, rewards [N] f32,
 * terminals/truncations/masks [N] u8. */
int pfa_squared_async_reset(void *state, const pfa_squared_config *cfg, int64_t seed, float *obs,
                            float *rewards, uint8_t *terminals, uint8_t *truncations, uint8_t *masks,
                            pfa_stream_t stream);
/* Pre-draw the targets of the next `rounds` reset rounds (N random.sample calls each, in env order)
 * from the shared stream into the tape.  The stream does not depend on actions, so this can run
 * ahead of the steps that consume it.  rounds <= tape_rounds - (rounds filled but not yet consumed). */
int pfa_squared_fill_tape(void *state, const pfa_squared_config *cfg, int32_t rounds, pfa_stream_t stream);
/* Serial.send(actions) (vector.py:137-156): done envs reset (action ignored), others step.
 * actions: int64 [N] on device. */
int pfa_squared_send(void *state, const pfa_squared_config *cfg, const int64_t *actions, float *obs,
                     float *rewards, uint8_t *terminals, uint8_t *truncations, uint8_t *masks,
                     pfa_stream_t stream);
/* Episode statistics of episodes finished since the last call with reset != 0
 * (what clean_pufferl.evaluate averages from infos, clean_pufferl.py:119-121,144-152):
 * out[0]=count, out[1]=sum episode_return, out[2]=sum episode_length, out[3]=sum score (f64, device); out[4] = the tape
 * underrun flag (non-zero if a reset ever needed a target round pfa_squared_fill_tape had not drawn yet) — `out4` holds 5 doubles. */
int pfa_squared_episode_stats(void *state, const pfa_squared_config *cfg, double *out4, int32_t reset,
                              pfa_stream_t stream);
/* Per-env view for infos of the LAST send: finished[N] u8, episode_return[N] f64, episode_length[N] i32,
 * score[N] f64 (valid where finished). */
int pfa_squared_last_infos(void *state, const pfa_squared_config *cfg, uint8_t *finished,
                           double *episode_return, int32_t *episode_length, double *score,
                           pfa_stream_t stream);
/* Introspection for tests: remaining target cells [N][nt] int32 (-1 = hit), stream word position. */
int pfa_squared_debug_targets(void *state, const pfa_squared_config *cfg, int32_t *cells, pfa_stream_t stream);
int pfa_squared_debug_stream_pos(void *state, const pfa_squared_config *cfg, uint64_t *pos_device, pfa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * MLP policy — pufferlib.models.Default (models.py:24-62) wrapped by frameworks.cleanrl.Policy
 * (cleanrl.py:50-66) with sample_logits (cleanrl.py:25-47): one Discrete head, or the MultiDiscrete branch (models.py:29-35,
 * 55-58: one decoder Linear per head; cleanrl.py:31-44: one multinomial per head, log-probabilities and entropies summed).
 * Flat fp32 parameter vector, in this order (obs_stride columns per encoder row, pad columns 0):
 *   encoder.weight [H][obs_stride], encoder.bias [H], decoder.weight [A][H], decoder.bias [A],
 *   value_head.weight [1][H], value_head.bias [1].
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t obs_dim;     /* true feature count (informational) */
    int32_t obs_stride;  /* floats per obs row: 16, 32, 64, 96 or 128; the recurrent entry points (pfa_lstm_*) also take 160 */
    int32_t hidden;      /* 128 */
    int32_t num_actions; /* A <= 15: decoder rows = logits of all heads */
    uint32_t heads;      /* 0: one Discrete(A) head.  MultiDiscrete: head h has (heads >> 4h) & 15 logits, heads back to back in
                          * decoder row order, sizes summing to A; actions are then stored packed the same way, head h's choice
                          * in bits 4h..4h+3.  The standalone policy steps (pfa_mlp_forward_sample, pfa_lstm_policy_step)
                          * and the updates (pfa_ppo_mlp_grad, pfa_lstm_heads_loss) take several heads; the fused device
                          * rollouts refuse them. */
} pfa_mlp_dims;

int64_t pfa_mlp_param_count(const pfa_mlp_dims *dims);

/* Philox4x32-10 action-noise stream: key=(seed lo, seed hi), counter=(row, column/4, step lo, step hi),
 * row = global env index (env_offset + local env), so the NOISE stream is invariant to how envs are sharded (env reset streams are per process, like the reference's). */
typedef struct {
    uint64_t seed;
    uint64_t step; /* rollout step counter (monotonic across rollouts) */
} pfa_noise_key;

/* policy(obs) in rollout mode: actions int64 [rows], logprob/entropy/value f32 [rows].
 * `noise` (nullable) is an explicit Exp(1) tensor [rows][A] (parity with torch.multinomial:
 * action = argmax(softmax(logits)/noise)); when null the Philox stream above is used with
 * row index = row_offset + r. */
int pfa_mlp_forward_sample(const float *obs, int64_t rows, const float *params, const pfa_mlp_dims *dims,
                           const float *noise, const pfa_noise_key *key, int64_t row_offset,
                           int64_t *actions, float *logprob, float *entropy, float *value,
                           pfa_stream_t stream);

/* The Exp(1) action noise of a whole rollout in one launch: out[t][r][j] (f32 [steps][rows][A]) = the draw the Philox
 * stream above makes for (row row_offset + r, step key->step + t, column j) — bit for bit what the policy steps draw in
 * place when their `noise` is null, so handing `out` to them as the explicit tensor changes nothing but where the
 * arithmetic runs (off the fused rollout's per-step critical path).  Stands in for torch.multinomial's internal
 * exponential_ draw (frameworks/cleanrl.py:25-47 sample_logits -> torch.multinomial). */
int pfa_philox_exp_noise(float *out, int64_t steps, int64_t rows, int32_t num_actions, const pfa_noise_key *key,
                         int64_t row_offset, pfa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Experience — clean_pufferl.Experience (clean_pufferl.py:380-482) kept ENV-MAJOR on device:
 * row (env e, step t) lives at flat index e*T + t, which is the order sort_training_data (:452-464)
 * produces for a Serial backend; no sort, no gather.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    float *obs;        /* [N*T][obs_stride] */
    int32_t *actions;  /* [N*T] */
    float *logprobs;   /* [N*T] */
    float *values;     /* [N*T] */
    float *rewards;    /* [N*T] */
    float *dones;      /* [N*T] (terminals as 0/1 floats, clean_pufferl.py:447) */
    float *advantages; /* [N*T] written by pfa_gae_f32 */
    float *returns;    /* [N*T] */
    int32_t horizon_T; /* steps per env per rollout */
} pfa_experience;

/* Fused rollout — clean_pufferl.evaluate (clean_pufferl.py:76-154) for a Squared vecenv and an MLP
 * policy: T x { recv, policy forward + sample, Experience.store, send }, all on device, envs resident
 * in LDS/registers for the whole rollout.  The tape must hold the reset rounds these T steps consume.
 * `noise` nullable [T][N][A]; env_offset = global index of local env 0 (rank * N).  On return the live
 * buffers hold what recv() would return next. */
int pfa_rollout_mlp_squared(void *state, const pfa_squared_config *cfg, const float *params,
                            const pfa_mlp_dims *dims, const pfa_experience *exp, const float *noise,
                            const pfa_noise_key *key, int64_t env_offset, float *obs, float *rewards,
                            uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream);

/* The same two kernels for a Default policy of another width — models.Default(env, hidden_size=H) (models.py:24-39) with H in
 * {64, 128, 256, 512}, the widths the reference's environment policies use (classic_control 64, nethack / nmmo 256, atari 512) —
 * reading the module's own tensors in torch's shapes through a view (no packed copy): encoder.weight [hidden][ldw1] (columns
 * >= obs_dim read as 0), encoder.bias, decoder.weight [num_actions][hidden], decoder.bias, value_head.weight [hidden],
 * value_head.bias.  One Discrete head of up to 15 actions; observation rows of obs_stride in {16, 32, 64} floats.  The policy
 * forward of the protocol path (pfa_mlp_view_forward_sample) and the persistent rollout (pfa_rollout_mlp_view_squared) run the
 * same tile code, so stepping through recv/send and the fused rollout give bit-identical numbers, as for the 128-wide policy.
 * pfa_mlp_view_supported: 1 when the shape is one these kernels take (other shapes: the GEMM path of general.py). */
typedef struct pfa_mlp_view {
    const float *w1;
    int32_t ldw1, obs_dim, obs_stride, hidden, num_actions, reserved;
    const float *b1, *w2, *b2, *wv, *bv;
} pfa_mlp_view;
int pfa_mlp_view_supported(const pfa_mlp_view *view);
int pfa_mlp_view_forward_sample(const float *obs, int64_t rows, const pfa_mlp_view *view, const float *noise,
                                const pfa_noise_key *key, int64_t row_offset, int64_t *actions, float *logprob,
                                float *entropy, float *value, pfa_stream_t stream);
int pfa_rollout_mlp_view_squared(void *state, const pfa_squared_config *cfg, const pfa_mlp_view *view, const pfa_experience *exp,
                                 const float *noise, const pfa_noise_key *key, int64_t env_offset, float *obs, float *rewards,
                                 uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream);



/* ------------------------------------------------------------------------------------------
 * Stochastic vecenv (a second ocean env family on device) — replaces pufferlib.vector.Serial
 * (vector.py:70-166) over ocean.environment.make_stochastic (ocean/environment.py:61-64; horizon 100) =
 * GymnasiumPufferEnv + EpisodeStats + ocean.Stochastic (ocean.py:529-582).  No randomness: state = (tick, count of
 * action 0); observation always [0.0] in rows of 16 floats; reward = f32(1 - (p - count/tick)^2) when the action moves
 * the action-0 fraction towards p, else 0.  `state` >= pfa_stochastic_state_bytes(num_envs); live buffers as Squared.
 * episode_stats: out4 = {episodes, sum return, sum length, sum score} (f64).
 * ------------------------------------------------------------------------------------------ */
size_t pfa_stochastic_state_bytes(int32_t num_envs);
int pfa_stochastic_async_reset(void *state, int32_t num_envs, float *obs, float *rewards, uint8_t *terminals,
                               uint8_t *truncations, uint8_t *masks, pfa_stream_t stream);
int pfa_stochastic_send(void *state, int32_t num_envs, double p, int32_t horizon, const int64_t *actions, float *obs,
                        float *rewards, uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream);
int pfa_stochastic_episode_stats(void *state, int32_t num_envs, double *out4, int32_t reset, pfa_stream_t stream);
int pfa_stochastic_last_infos(void *state, int32_t num_envs, uint8_t *finished, double *episode_return,
                              int32_t *episode_length, double *score, pfa_stream_t stream);
/* clean_pufferl.evaluate's loop for a Stochastic vecenv and the MLP policy, one persistent kernel (as
 * pfa_rollout_mlp_squared). */
int pfa_rollout_mlp_stochastic(void *state, int32_t num_envs, double p, int32_t horizon, const float *params,
                               const pfa_mlp_dims *dims, const pfa_experience *exp, const float *noise,
                               const pfa_noise_key *key, int64_t env_offset, float *obs, float *rewards,
                               uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Memory vecenv (the env family that needs the recurrent policy) — replaces pufferlib.vector.Serial over
 * ocean.environment.make_memory (ocean/environment.py:41-44) = GymnasiumPufferEnv + EpisodeStats + ocean.Memory
 * (ocean.py:65-123).  Every reset draws np.random.randint(0, 2, size=2L+D) from numpy's process-global legacy MT19937
 * (seeded per env at async_reset, shared afterwards); the stream is action-independent and is drawn ahead into a tape
 * (pfa_memory_fill_tape) like Squared's.  Observation rows of 16 floats (1 real column), 2 actions.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t num_envs;
    int32_t mem_length;  /* L, 1..16 */
    int32_t mem_delay;   /* D >= 0; horizon = 2L + D */
    int32_t tape_rounds; /* ring capacity in reset rounds */
} pfa_memory_config;
size_t pfa_memory_state_bytes(const pfa_memory_config *cfg);
int pfa_memory_async_reset(void *state, const pfa_memory_config *cfg, int64_t seed, float *obs, float *rewards,
                           uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream);
int pfa_memory_fill_tape(void *state, const pfa_memory_config *cfg, int32_t rounds, pfa_stream_t stream);
int pfa_memory_send(void *state, const pfa_memory_config *cfg, const int64_t *actions, float *obs, float *rewards,
                    uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream);
/* out4[0..3] as Squared's; out4[4] = tape underrun flag (5 doubles) */
int pfa_memory_episode_stats(void *state, const pfa_memory_config *cfg, double *out4, int32_t reset, pfa_stream_t stream);
int pfa_memory_last_infos(void *state, const pfa_memory_config *cfg, uint8_t *finished, double *episode_return,
                          int32_t *episode_length, double *score, pfa_stream_t stream);
/* test introspection: bit j of bits[e] = solution[j] of env e; *underrun != 0 if a reset ever found the tape empty */
int pfa_memory_debug_solutions(void *state, const pfa_memory_config *cfg, uint32_t *bits, int32_t *underrun,
                               pfa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Bandit vecenv — replaces pufferlib.vector.Serial over ocean.environment.make_bandit
 * (ocean/environment.py:33-37) = GymnasiumPufferEnv + EpisodeStats + ocean.Bandit (ocean.py:8-63).  Every episode is
 * one step; reward = f32(((action == solution) + noise[e]) * reward_scale) in f64 arithmetic.  `solution` and the per-env
 * `noise` table (f64 [num_envs], device; NULL = reward_noise 0) are what numpy's legacy generator yields after
 * seed(42): the host side draws them with numpy itself.  Observation rows of 16 floats, first column 1.0.
 * ------------------------------------------------------------------------------------------ */
size_t pfa_bandit_state_bytes(int32_t num_envs);
int pfa_bandit_async_reset(void *state, int32_t num_envs, float *obs, float *rewards, uint8_t *terminals,
                           uint8_t *truncations, uint8_t *masks, pfa_stream_t stream);
int pfa_bandit_send(void *state, int32_t num_envs, int32_t solution, double reward_scale, const double *noise,
                    const int64_t *actions, float *obs, float *rewards, uint8_t *terminals, uint8_t *truncations,
                    uint8_t *masks, pfa_stream_t stream);
int pfa_bandit_episode_stats(void *state, int32_t num_envs, double *out4, int32_t reset, pfa_stream_t stream);
int pfa_bandit_last_infos(void *state, int32_t num_envs, uint8_t *finished, double *episode_return,
                          int32_t *episode_length, double *score, pfa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Multiagent vecenv — replaces pufferlib.vector.Serial over ocean.environment.make_multiagent
 * (ocean/environment.py:76-79) = PettingZooPufferEnv + MultiagentEpisodeStats + ocean.Multiagent (ocean.py:148-224): two
 * agent rows per env (row 2e: observation 0, scores with action 0; row 2e+1: observation 1, scores with action 1), every
 * step terminal, the next send the reset row.  All row arrays hold 2 * num_envs rows (observations 16 floats per row).
 * episode_stats: out4 = (finished steps of agent 1, its score sum, the same for agent 2) — the per-slot means are the
 * `1/score`, `2/score` statistics clean_pufferl.evaluate derives from the reference's infos.
 * ------------------------------------------------------------------------------------------ */
size_t pfa_multiagent_state_bytes(int32_t num_envs);
int pfa_multiagent_async_reset(void *state, int32_t num_envs, float *obs, float *rewards, uint8_t *terminals,
                               uint8_t *truncations, uint8_t *masks, pfa_stream_t stream);
int pfa_multiagent_send(void *state, int32_t num_envs, const int64_t *actions, float *obs, float *rewards,
                        uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream);
int pfa_multiagent_episode_stats(void *state, int32_t num_envs, double *out4, int32_t reset, pfa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Device-resident ocean `Spaces` vecenv — replaces pufferlib.vector.Serial over make_spaces
 * (ocean/environment.py:66-69) = GymnasiumPufferEnv + EpisodeStats + ocean.Spaces (ocean.py:356-404): Dict observation
 * emulated into 108-byte rows {flat int8[5] @0, image f32[5][5] @8}, handed to the policy as rows of 128 floats holding the
 * BYTE values (what models.Default computes with observations.float(), models.py:50); Dict action = MultiDiscrete([2, 2]),
 * here one packed word per env (head 0 = flat in bits 0..3, head 1 = image in bits 4..7, the pfa_mlp_dims.heads convention).
 * Every step is terminal, the next send the reset row.  Observations come from numpy's process-global legacy generator
 * (randn(5,5) then randint(-1, 2, 5, int8) per reset, in env order): pfa_spaces_async_reset(seed) = np.random.seed(seed)
 * (what clean_pufferl.seed_everything does, clean_pufferl.py:596-600) + the N initial resets; later resets read the tape that
 * pfa_spaces_fill_tape draws ahead (`rounds` x N resets; csrc/spaces.hip resolves the data-dependent stream positions in parallel).
 * episode_stats: out[0..3] = {episodes, sum return, sum length, sum score}, out[4] = tape underrun flag (5 doubles).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t num_envs;
    int32_t tape_rounds; /* ring capacity in reset rounds (>= 2) */
} pfa_spaces_config;
size_t pfa_spaces_state_bytes(const pfa_spaces_config *cfg);
int pfa_spaces_async_reset(void *state, const pfa_spaces_config *cfg, int64_t seed, float *obs, float *rewards,
                           uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream);
int pfa_spaces_fill_tape(void *state, const pfa_spaces_config *cfg, int32_t rounds, pfa_stream_t stream);
int pfa_spaces_send(void *state, const pfa_spaces_config *cfg, const int64_t *actions, float *obs, float *rewards,
                    uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream);
int pfa_spaces_episode_stats(void *state, const pfa_spaces_config *cfg, double *out5, int32_t reset, pfa_stream_t stream);
int pfa_spaces_last_infos(void *state, const pfa_spaces_config *cfg, uint8_t *finished, double *episode_return,
                          int32_t *episode_length, double *score, pfa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Synthetic byte-row vecenv — the device-side generator of BASELINE configs[2]'s workload SHAPE:
 * MiniGrid-shaped rows (minigrid/environment.py:14-48: 160 emulated bytes, 7 actions, episodes cut at 100 steps) whose
 * arithmetic lives in the third-party `minigrid` package, not in the reference tree — env parity is unpinned by construction,
 * so a counter-based generator (csrc/synth_env.hpp) supplies observations and a learnable reward; what is held to parity on
 * this workload is the policy / GAE / PPO side.  Protocol and buffers as the other device envs; observations are rows of
 * `obs_stride` floats holding `obs_values` byte values in 0..obs_high.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t num_envs;
    int32_t obs_values;     /* values per row (<= 160) */
    int32_t obs_stride;     /* floats per row, multiple of 16 */
    int32_t num_actions;    /* 2..15 */
    int32_t episode_length; /* steps per episode; the send after the terminal row is the reset row */
    int32_t obs_high;       /* values are uniform in 0..obs_high */
    uint64_t seed;
    int64_t env_offset;     /* global index of local env 0 (sharding) */
} pfa_synth_config;
size_t pfa_synth_state_bytes(const pfa_synth_config *cfg);
int pfa_synth_async_reset(void *state, const pfa_synth_config *cfg, float *obs, float *rewards, uint8_t *terminals,
                          uint8_t *truncations, uint8_t *masks, pfa_stream_t stream);
int pfa_synth_send(void *state, const pfa_synth_config *cfg, const int64_t *actions, float *obs, float *rewards,
                   uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream);
/* Frame rows (BASELINE configs[3]: uint8 (framestack, 84, 84) observations, SURVEY config C4): obs_values = obs_stride = bytes
 * per row (a multiple of 16), obs_high = 255; same generator, state and statistics entries, uint8 live buffer. */
int pfa_frames_async_reset(void *state, const pfa_synth_config *cfg, uint8_t *obs, float *rewards, uint8_t *terminals,
                           uint8_t *truncations, uint8_t *masks, pfa_stream_t stream);
int pfa_frames_send(void *state, const pfa_synth_config *cfg, const int64_t *actions, uint8_t *obs, float *rewards,
                    uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream);
int pfa_synth_episode_stats(void *state, const pfa_synth_config *cfg, double *out4, int32_t reset, pfa_stream_t stream);
int pfa_synth_last_infos(void *state, const pfa_synth_config *cfg, uint8_t *finished, double *episode_return,
                         int32_t *episode_length, double *score, pfa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Structured-observation unpack — replaces pufferlib.pytorch.nativize_tensor
 * (pufferlib/pytorch.py:96-145): flat emulated rows [num_rows][row_bytes] (device, 16-byte aligned) -> one dense tensor
 * per leaf of the observation space, in one launch.  A field is a leaf as pufferlib.pytorch.nativize_dtype
 * (pytorch.py:48-94) describes it: byte offset inside the row, element count, element dtype.  to_f32 == 0 copies the
 * elements as they are into `out` ([num_rows][count], dense, 16-byte aligned); to_f32 == 1 converts them to f32
 * (the `.float()` encoders apply next, models.py:50) and writes rows `out_stride` floats apart, so leaves can land in
 * column ranges of one [num_rows][total] matrix (torch.cat of the flattened leaves).
 * ------------------------------------------------------------------------------------------ */
#define PFA_NAT_MAX_FIELDS 32
typedef enum {
    PFA_NAT_U8 = 0, PFA_NAT_I8 = 1, PFA_NAT_U16 = 2, PFA_NAT_I16 = 3, PFA_NAT_U32 = 4, PFA_NAT_I32 = 5,
    PFA_NAT_U64 = 6, PFA_NAT_I64 = 7, PFA_NAT_F16 = 8, PFA_NAT_F32 = 9, PFA_NAT_F64 = 10
} pfa_nat_dtype;
typedef struct {
    void *out;          /* device */
    int32_t offset;     /* bytes from the start of a row */
    int32_t count;      /* elements per row */
    int32_t dtype;      /* pfa_nat_dtype of the source elements */
    int32_t to_f32;
    int32_t out_stride; /* elements between output rows; == count unless to_f32 */
    int32_t reserved;
} pfa_nat_field;
int pfa_nativize_rows(const void *rows, int64_t num_rows, int32_t row_bytes, const pfa_nat_field *fields /* host */,
                      int32_t num_fields, pfa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * PPO update — the minibatch loop of clean_pufferl.train (clean_pufferl.py:175-258).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    float clip_coef, vf_clip_coef, vf_coef, ent_coef;
    int32_t norm_adv, clip_vloss;
    int32_t num_minibatches; /* nmb */
    int32_t bptt_horizon;    /* segment length; minibatch m = segments {m + k*nmb} (:455-457) */
} pfa_ppo_hparams;

/* The workspace must be allocated ZERO-INITIALISED once (its last region holds the {value, generation} hand-off words of
 * pfa_ppo_mlp_train's one-launch reduce + Adam: nothing else writes there and the library never clears it). */
size_t pfa_ppo_workspace_bytes(const pfa_mlp_dims *dims, int64_t batch_rows, const pfa_ppo_hparams *hp);
/* Per-minibatch advantage statistics (sum, sum of squares in f64; count) for norm_adv (:211-213).
 * stats: f64 [nmb][2] on device.  Call once per update after GAE; with several ranks, all-reduce(sum)
 * `stats` and pass the global row count per minibatch to the update. */
int pfa_ppo_adv_stats(const pfa_experience *exp, int64_t batch_rows, const pfa_ppo_hparams *hp,
                      double *stats, void *workspace, pfa_stream_t stream);
/* Forward + loss + backward for minibatch `mb`: writes the flat gradient (same layout as params, already
 * divided by `global_mb_rows`, so an all-reduce SUM over ranks yields the global-minibatch mean gradient)
 * followed by 16 floats: the loss sums over this rank's rows, accumulated in f64 and stored as (hi, lo) float pairs
 * (sum_i = (double)grads[P+2i] + (double)grads[P+2i+1]; an f32 all-reduce of the bucket keeps ~48 bits of each sum):
 *   i = 0..5: sum pg_loss, sum v_loss, sum entropy, sum -logratio, sum (ratio-1-logratio), sum clipped.
 */
int pfa_ppo_mlp_grad(const pfa_experience *exp, int64_t batch_rows, int32_t mb, const float *params,
                     const pfa_mlp_dims *dims, const pfa_ppo_hparams *hp, const double *adv_stats,
                     int64_t global_mb_rows, float *grads, void *workspace, pfa_stream_t stream);
/* MFMA instructions (v_mfma_f32_16x16x4_f32, 2048 FLOP each) the gradient kernel instantiated for these dimensions issues per
 * 16-row tile: what bench.py's roofline.frac_executed is computed from.  0 = unsupported dimensions. */
int pfa_ppo_mlp_grad_mfma_per_tile(int32_t obs_dim, int32_t obs_stride, int32_t num_actions);
/* Which kernel pfa_ppo_mlp_grad / pfa_ppo_mlp_train launch for these dimensions and local minibatch rows under the current product
 * form (pfa_igemm_set_products): 0 = ppo_mlp_grad_kernel (exact fp32 MFMA chains), 1 = ppo_mlp_grad_bf16_kernel (csrc/ppo_bf16.hpp:
 * the opt-in form, the 7x7 grid on 64-float rows with minibatches of whole 32-row tiles; everything else stays on 0). */
/* pfa_ppo_mlp_train's reduce + Adam launch hands the clip norm's pieces over grid-wide; the wait is bounded (PFA_WAIT_TIMEOUT_MS,
 * default 10 000) and the one-launch form is only used when the runtime's occupancy says all its workgroups are resident at once.
 * 0 = ok, 1 = a wait ran out (the parameters hold NaN since).  A plain host read, no synchronisation. */
int pfa_ppo_grid_status(void);
/* Clears the word once it has been reported (returns the value it held), so that the recovery the error message names — PFA_FUSED_ADAM=0
 * and restored parameters — can happen inside the same process. */
int pfa_ppo_grid_reset(void);
int pfa_ppo_mlp_grad_path(const pfa_mlp_dims *dims, int64_t mb_rows);
/* clip_grad_norm_(max_grad_norm) + Adam(eps) step (:240-244; torch.optim.Adam single-tensor semantics, bias
 * correction with `step` = 1-based optimizer step count).  grad_scale multiplies grads first.
 * If loss_sums (the 16-float tail above) / losses (f64[8]) are given: losses[i] += sum_i * loss_scale for i < 6, i.e. the running
 * {policy_loss, value_loss, entropy, old_approx_kl, approx_kl, clipfrac} of clean_pufferl.py:249-254 with
 * loss_scale = 1 / (global_mb_rows * num_minibatches).  norm_partials (nullable): n f64 pieces of sum(g^2) left in
 * the workspace by pfa_ppo_mlp_grad (valid only when grads were not modified since, i.e. single rank); when
 * null the norm is recomputed from `grads` (after an all-reduce). */
int pfa_adam_clip_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t count,
                       float lr, float beta1, float beta2, float eps, int64_t step, float max_grad_norm,
                       float grad_scale, const float *loss_sums, double *losses, double loss_scale,
                       const double *norm_partials, int32_t n_norm_partials, pfa_stream_t stream);

/* The whole minibatch loop of one clean_pufferl.train call on ONE rank (clean_pufferl.py:175-258 without the
 * target_kl early exit): update_epochs x num_minibatches x { pfa_ppo_mlp_grad, pfa_adam_clip_step } enqueued from
 * native code (no per-step host work).  opt_step = optimizer steps taken before this call; losses as above.
 * data_parallel != 0: every optimizer step sums the flat bucket [gradient | 16 loss-sum floats] over the ranks (gradients are
 * pre-divided by the GLOBAL minibatch rows = local rows x world size; adv_stats must already hold the all-reduced sums) and takes
 * the clip norm after the sum.  Transport: with the peer path open (pfa_p2p_open) the exchange runs inside the reduce + Adam
 * launch itself (two launches per optimizer step, PFA_FUSED_DP=0 turns that off); otherwise one all-reduce per step through
 * pfa_dist_all_reduce_f32 (peer path as a launch of its own, else the RCCL communicator of pfa_dist_init) on `stream`.
 * Single rank: sum of the partials, clip norm and Adam are one launch as well (PFA_FUSED_ADAM=0: the two-kernel form of
 * pfa_ppo_mlp_grad + pfa_adam_clip_step, bit-identical results).  The one-launch form hands the pieces of the clip norm from
 * workgroup to workgroup through {value, launch generation} words in a small LIBRARY-OWNED device buffer (one per workspace pointer,
 * allocated and cleared on first use): the workspace itself needs no initialisation and may be recycled freely. */
int pfa_ppo_mlp_train(const pfa_experience *exp, int64_t batch_rows, float *params, const pfa_mlp_dims *dims,
                      const pfa_ppo_hparams *hp, const double *adv_stats, float *grads, float *exp_avg,
                      float *exp_avg_sq, int64_t opt_step, float lr, float beta1, float beta2, float eps,
                      float max_grad_norm, int32_t update_epochs, double *losses, void *workspace,
                      int32_t data_parallel, pfa_stream_t stream);
/* pfa_ppo_mlp_train + the report clean_pufferl.train ends with (clean_pufferl.py:249-254, 266-270): when the update runs in the
 * one-launch form, its LAST reduce + Adam launch also writes log_out10 = { losses[0..5], log_ev4[0..3] } — what
 * pfa_train_log_pack leaves, without a launch of its own behind the update (the report is what the host waits for).
 * *log_packed (host memory, may be NULL) = 1 when it did, 0 when the caller still has to call pfa_train_log_pack (two-kernel
 * form, all-reduce as its own step, update_epochs == 0).  log_ev4 / log_out10 NULL: exactly pfa_ppo_mlp_train.
 * log_out10 may be device-visible pinned host memory. */
int pfa_ppo_mlp_train_logged(const pfa_experience *exp, int64_t batch_rows, float *params, const pfa_mlp_dims *dims,
                             const pfa_ppo_hparams *hp, const double *adv_stats, float *grads, float *exp_avg,
                             float *exp_avg_sq, int64_t opt_step, float lr, float beta1, float beta2, float eps,
                             float max_grad_norm, int32_t update_epochs, double *losses, void *workspace,
                             int32_t data_parallel, const double *log_ev4, double *log_out10, int32_t *log_packed,
                             pfa_stream_t stream);

/* The minibatch step of clean_pufferl.train (clean_pufferl.py:175-244 up to loss.backward()) for a pfa_mlp_view policy (declared with the rollout kernels above) with hidden in
 * {64, 256, 512} (csrc/ppo_wide.hip: the hidden dimension split over the four wavefronts of a workgroup, every wave's slice of the
 * weights and of every gradient in its registers for the whole launch): forward + sample_logits with the stored actions + PPO loss
 * + backward over minibatch `mb` — the counterpart of pfa_ppo_mlp_grad for the 128-wide kernel-layout policy.  `grads` names the
 * same six tensors inside the gradient buffer (torch shapes: what loss.backward() leaves in .grad, already divided by
 * global_mb_rows), tail16 receives the loss sums as in pfa_ppo_mlp_grad.  workspace: pfa_ppo_wide_workspace_bytes(params).
 * Clip + Adam: pfa_adam_clip_step on the flat buffer.  pfa_ppo_wide_supported: 1 when the shape is one this kernel takes. */
int pfa_ppo_wide_supported(const pfa_mlp_view *view);
size_t pfa_ppo_wide_workspace_bytes(const pfa_mlp_view *view);
int pfa_ppo_wide_grad(const pfa_experience *exp, int64_t batch_rows, int32_t mb, const pfa_mlp_view *params, const pfa_mlp_view *grads,
                      float *tail16, const pfa_ppo_hparams *hp, const double *adv_stats, int64_t global_mb_rows, void *workspace,
                      pfa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Recurrent policy — pufferlib.models.LSTMWrapper (models.py:64-111): Default encoder -> nn.LSTM(128,128,1) ->
 * Default heads, gate order i,f,g,o.  Flat parameter vector = the MLP vector above followed by
 *   recurrent.weight_ih_l0 [512][128], weight_hh_l0 [512][128], bias_ih_l0 [512], bias_hh_l0 [512].
 * Every product runs on the fused MFMA kernels declared further down (pfa_lstm_pack / pfa_rollout_lstm_squared /
 * pfa_lstm_policy_step / pfa_lstm_seq_forward / pfa_lstm_seq_backward / pfa_gemm_tn_f32).  All row-indexed buffers are
 * dense [rows][...] fp32.
 * ------------------------------------------------------------------------------------------ */
/* --- fused recurrent policy (csrc/lstm_fused.hip) ---
 * Flat parameter vector of the recurrent policy = the MLP block above followed by nn.LSTM's weight_ih_l0 [512][128],
 * weight_hh_l0 [512][128], bias_ih_l0 [512], bias_hh_l0 [512].  pfa_lstm_pack re-tiles [W_ih | W_hh] into the MFMA
 * fragment order the fused kernels stream from L2 (wpack: pfa_lstm_pack_bytes() bytes, 16-byte aligned); call it
 * whenever the weights changed. */
int64_t pfa_lstm_param_count(const pfa_mlp_dims *dims);
size_t pfa_lstm_pack_bytes(void);
int pfa_lstm_pack(const float *params, const pfa_mlp_dims *dims, void *wpack, pfa_stream_t stream);
/* RecurrentPolicy.forward with action=None (frameworks/cleanrl.py:84-93 -> models.py:86-111) on `rows` rows: encode ->
 * one nn.LSTM step -> decode -> sample_logits.  h, c [rows][128] are the LSTM state, updated in place.  Outputs as in
 * pfa_mlp_forward_sample. */
int pfa_lstm_policy_step(const float *obs, int64_t rows, const float *params, const pfa_mlp_dims *dims,
                         const void *wpack, float *h, float *c, const float *noise, const pfa_noise_key *key,
                         int64_t row_offset, int64_t *actions, float *logprob, float *entropy, float *value,
                         pfa_stream_t stream);
/* clean_pufferl.evaluate's loop (clean_pufferl.py:84-124) for a Squared vecenv and the recurrent policy, one persistent
 * kernel: arguments as pfa_rollout_mlp_squared plus the packed gate weights and the LSTM state h, c [num_envs][128]
 * (Experience.lstm_h / lstm_c, clean_pufferl.py:407-412), carried across rollouts and never reset on done. */
int pfa_rollout_lstm_squared(void *state, const pfa_squared_config *cfg, const float *params,
                             const pfa_mlp_dims *dims, const void *wpack, float *h, float *c,
                             const pfa_experience *exp, const float *noise, const pfa_noise_key *key,
                             int64_t env_offset, float *obs, float *rewards, uint8_t *terminals,
                             uint8_t *truncations, uint8_t *masks, pfa_stream_t stream);
/* The same persistent rollout over the ocean Memory vecenv (pfa_memory_*): clean_pufferl.evaluate (clean_pufferl.py:76-154) for
 * RecurrentPolicy(LSTMWrapper(Default)) on make_memory envs as ONE launch; the solution tape must hold the reset rounds of
 * these horizon_T sends (pfa_memory_fill_tape). */
int pfa_rollout_lstm_memory(void *state, const pfa_memory_config *cfg, const float *params, const pfa_mlp_dims *dims,
                            const void *wpack, float *h, float *c, const pfa_experience *exp, const float *noise,
                            const pfa_noise_key *key, int64_t env_offset, float *obs, float *rewards, uint8_t *terminals,
                            uint8_t *truncations, uint8_t *masks, pfa_stream_t stream);
/* ... and over the synthetic byte-row vecenv (pfa_synth_*): BASELINE configs[2]'s rollout as one launch. */
int pfa_rollout_lstm_synth(void *state, const pfa_synth_config *cfg, const float *params, const pfa_mlp_dims *dims,
                           const void *wpack, float *h, float *c, const pfa_experience *exp, const float *noise,
                           const pfa_noise_key *key, int64_t env_offset, float *obs, float *rewards, uint8_t *terminals,
                           uint8_t *truncations, uint8_t *masks, pfa_stream_t stream);
/* --- the recurrent policy in training mode (csrc/lstm_seq.hip): forward over the `steps` (= bptt_horizon) time steps
 * of a minibatch of `rows` independent segments and back-propagation through time (clean_pufferl.py:186-193, :244).
 * Time-major buffers: obs_tm [steps][rows][obs_stride], xe / dh_heads / dxe [steps][rows][128], gates_act / dgates
 * [steps][rows][512], hs / cs [steps+1][rows][128] (slot 0 = the carried-in state, read; slots 1.. written).
 * forward keeps what autograd would keep (xe, gate activations i,f,g,o, c_t, h_t).  backward takes d loss / d h_t through
 * the heads (dh_heads, from pfa_lstm_heads_loss) and produces dgates (d loss / d gate pre-activations), dxe (d loss / d the
 * encoder pre-activations, relu' applied) and the bias gradients gate_bias_grad [512] (= d b_ih = d b_hh) and
 * enc_bias_grad [128]; wpack_bwd from pfa_lstm_pack_bwd (same size as wpack; pfa_lstm_pack_both: both re-tilings in one launch).
 * init_slot: where the carried-in state is read — 0: slot 0 of hs / cs as it is; k in 1..steps: slot k (the previous minibatch's final
 * state, clean_pufferl.py:188-191), copied into slot 0 by the kernel; < 0: zero state (lstm_state = None, :176), slot 0 zeroed.
 * backward with gate_bias_grad = enc_bias_grad = NULL leaves the per-workgroup column sums in `workspace` ([ceil(rows / 32)][640])
 * for the caller's own reduction (pfa_reduce_multi, kind 1). */
int pfa_lstm_pack_bwd(const float *params, const pfa_mlp_dims *dims, void *wpack_bwd, pfa_stream_t stream);
int pfa_lstm_pack_both(const float *params, const pfa_mlp_dims *dims, void *wpack, void *wpack_bwd, pfa_stream_t stream);
int pfa_lstm_seq_forward(const float *obs_tm, int64_t rows, int32_t steps, const float *params, const pfa_mlp_dims *dims,
                         const void *wpack, float *xe, float *gates_act, float *hs, float *cs, int32_t init_slot, pfa_stream_t stream);
size_t pfa_lstm_seq_backward_workspace_bytes(int64_t rows);
int pfa_lstm_seq_backward(const float *gates_act, const float *cs, const float *xe, const float *dh_heads, int64_t rows,
                          int32_t steps, const void *wpack_bwd, float *dgates, float *dxe, float *gate_bias_grad,
                          float *enc_bias_grad, void *workspace, pfa_stream_t stream);
/* Experience.store of rollout step t (clean_pufferl.py:436-450) into the env-major buffers. */
int pfa_store_step(const pfa_experience *exp, int32_t t, int32_t num_envs, int32_t obs_stride, const float *obs,
                   const float *rewards, const uint8_t *terminals, const int64_t *actions, const float *logprob,
                   const float *value, pfa_stream_t stream);
/* Experience.store for rows arriving in ANY order from a host vecenv (clean_pufferl.py:436-450), fused with
 * sort_training_data (:452-464): batch row i (mask[i] != 0; mask nullable = all) of env slot env_ids[i] (nullable = i; unique
 * within a call; 0 <= id < num_slots) is written to its sorted position env_id*T + counters[env_id]++ of the env-major
 * buffers.  counters: int32 [num_slots] rows held per env (zero it when a rollout starts); stored_dropped: int32 [2]
 * running totals of rows stored / rows dropped because their env already held T rows or the id was out of range. */
int pfa_store_rows(const pfa_experience *exp, int32_t rows, int32_t num_slots, int32_t obs_stride, const float *obs,
                   const float *rewards, const uint8_t *dones, const int64_t *actions, const float *logprob,
                   const float *value, const int32_t *env_ids, const uint8_t *mask, int32_t *counters,
                   int32_t *stored_dropped, pfa_stream_t stream);
/* Observation rows of minibatch mb in TIME-MAJOR order (row t*R + k = segment mb + k*nmb, step t). */
int pfa_gather_obs_time_major(const pfa_experience *exp, int64_t batch_rows, int32_t mb, const pfa_ppo_hparams *hp,
                              int32_t obs_stride, float *out, pfa_stream_t stream);
/* decode_actions + PPO loss on the time-major hidden states h [mbs][128] of minibatch mb: dout [mbs][16] (d loss / d the
 * padded head outputs), dh [mbs][128] (d loss / d h through the heads), loss_sums8 = the 16-float (hi, lo) tail of pfa_ppo_mlp_grad,
 * head_bias_grad16 (nullable) = column sums of dout = d loss / d (decoder.bias | value_head.bias | padding). */
size_t pfa_lstm_heads_loss_workspace_bytes(void);
int pfa_lstm_heads_loss(const float *h, const pfa_experience *exp, int64_t batch_rows, int32_t mb, const float *params,
                        const pfa_mlp_dims *dims, const pfa_ppo_hparams *hp, const double *adv_stats,
                        int64_t global_mb_rows, float *dout, float *dh, float *loss_sums8, float *head_bias_grad16,
                        void *workspace, pfa_stream_t stream);
/* C[mo][no] (row stride ldc) = sum over the k rows of A[k][mo]^T B[k][no]  (A, B row-major activations with row strides
 * lda, ldb; fp32 MFMA, split over k, deterministic f64 reduction of the splits).  These are the weight-gradient
 * contractions autograd performs for nn.Linear / nn.LSTM under loss.backward() (clean_pufferl.py:244).  Supported
 * (mo, no): mo a multiple of 128 with no a multiple of 16, or mo a multiple of 16 with no a multiple of 128.
 * workspace >= pfa_gemm_tn_workspace_bytes (0 = unsupported). */
size_t pfa_gemm_tn_workspace_bytes(int32_t mo, int32_t no, int64_t k);
int pfa_gemm_tn_f32(const float *a, int64_t lda, const float *b, int64_t ldb, float *c, int64_t ldc, int32_t mo,
                    int32_t no, int64_t k, void *workspace, pfa_stream_t stream);
/* Two products that share their first operand in ONE pass over it: c0[mo][128] = a^T b0, c1[mo][128] = a^T b1 (mo a multiple of
 * 128) — the recurrent layer's dW_ih = dG^T xe and dW_hh = dG^T h_prev (autograd's two matmuls in nn.LSTM's backward,
 * clean_pufferl.py:244): dG is fetched and staged once instead of twice.  workspace >= pfa_gemm_tn2_workspace_bytes. */
size_t pfa_gemm_tn2_workspace_bytes(int32_t mo, int64_t k);
/* The same products as their split partials only, each handing back the reduction that finishes it, and up to six such reductions in
 * ONE launch (the recurrent update's three weight-gradient products + its bias column sums were four latency-bound launches in a row).
 *   kind 0  c[i][j] (j < nb or c2 == NULL) / c2[i][j - nb] <- f64 fixed-order sum over `splits` partials [mo][no]
 *   kind 1  c[col] (col < nb) / c2[col - nb] <- column sums of `splits` rows of `no` floats (per-workgroup partial sums)
 * pfa_gemm_tn*_partial_f32 fill in everything of `job` but c / ldc / c2 / ldc2 (tn2 sets nb = 128). */
typedef struct pfa_reduce_job {
    int32_t kind, splits, mo, no;
    const float *partial;
    float *c;
    int64_t ldc;
    float *c2;
    int64_t ldc2;
    int32_t nb, reserved;
} pfa_reduce_job;
int pfa_gemm_tn_partial_f32(const float *a, int64_t lda, const float *b, int64_t ldb, int32_t mo, int32_t no, int64_t k, void *workspace,
                            pfa_reduce_job *job, pfa_stream_t stream);
int pfa_gemm_tn2_partial_f32(const float *a, int64_t lda, const float *b0, int64_t ldb0, const float *b1, int64_t ldb1, int32_t mo, int64_t k,
                             void *workspace, pfa_reduce_job *job, pfa_stream_t stream);
int pfa_reduce_multi(const pfa_reduce_job *jobs, int32_t njobs, pfa_stream_t stream);
int pfa_gemm_tn2_f32(const float *a, int64_t lda, const float *b0, int64_t ldb0, const float *b1, int64_t ldb1, float *c0, int64_t ldc0,
                     float *c1, int64_t ldc1, int32_t mo, int64_t k, void *workspace, pfa_stream_t stream);
/* Scatter into the flat gradient vector (MLP block + LSTM block layout): g16 [16][128] = dout^T h (rows < A ->
 * decoder.weight, row A -> value_head.weight), bsum16 = column sums of dout (-> decoder.bias, value_head.bias), and
 * d bias_hh_l0 = d bias_ih_l0 (already in place). */
int pfa_lstm_finish_grads(float *grads, const pfa_mlp_dims *dims, const float *g16, const float *bsum16, pfa_stream_t stream);
/* n f64 pieces of sum(g^2) over `count` gradient entries, for pfa_adam_clip_step's norm_partials. */
int pfa_sumsq_partials(const float *grads, int64_t count, double *partials, int32_t n, pfa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Data parallel over the GPUs of one node (no reference counterpart): one process per GPU, one RCCL communicator
 * per process, bound at run time (dlopen librccl.so.1).  Rank 0 creates the 128-byte id, the host side broadcasts it
 * (torch.distributed), every rank calls pfa_dist_init collectively.  Collectives run on the caller's stream.
 * ------------------------------------------------------------------------------------------ */
int pfa_dist_unique_id(uint8_t *id128_host);
int pfa_dist_init(const uint8_t *id128_host, int32_t rank, int32_t world);
int pfa_dist_finalize(void);
int pfa_dist_all_reduce_f32(float *buf, int64_t count, pfa_stream_t stream);   /* one-shot peer path when it is open and the bucket fits, else RCCL */
int pfa_dist_all_reduce_f64(double *buf, int64_t count, pfa_stream_t stream);
/* out8: [0] RCCL communicator up, [1] its ncclCommCount, [2] peer path open, [3] its world size, [4] its slot capacity (bytes),
 * [5] all-reduces sent over the peer path, [6] all-reduces sent over RCCL, [7] pfa_p2p_status. */
int pfa_dist_info(int64_t *out8);

/* ------------------------------------------------------------------------------------------
 * NatureCNN policy of BASELINE configs[3] — pufferlib.models.Convolutional (models.py:113-157) behind frameworks.cleanrl.Policy:
 * Conv2d(F,32,8,s4) ReLU Conv2d(32,64,4,s2) ReLU Conv2d(64,64,3,s1) ReLU Flatten Linear(3136,512) ReLU, actor Linear(512,A),
 * value_fn Linear(512,1); observations uint8 (F,84,84), `.float() / 255.0`.  Every product of forward and backward is a launch
 * of one of two fp32-MFMA implicit-GEMM kernels (csrc/igemm.hip); the operand A(m, k) is an access pattern, not a buffer:
 *   mode 0 dense            A = ptr[m * lda + k]
 *   mode 1 im2col, f32      ptr = NHWC activations [n][IH][IW][IC]; m = (n, oy, ox), k = (ky*KW + kx)*IC + ic
 *   mode 2 im2col, uint8    ptr = NCHW frames [n][IC][IH][IW];      m = (n, oy, ox), k = (ic*KH + ky)*KW + kx, the byte as a float 0..255
 *                           (`observations / 255.0`, models.py:150, rides in the other operand: pack_conv / perm 3 below)
 *   mode 3 col2im (for dX)  ptr = NHWC dOut [n][OH][OW][OC]; m = INPUT pixel (n, y, x); runs as S*S phases (y mod S, x mod S), each
 *                           contracting over the (KH/S)(KW/S) taps that reach it: k = (jy*(KW/S) + jx)*OC + oc, tap (py + jy*S, px + jx*S)
 * pfa_igemm_rows:    C[m][n] = epilogue(sum_k A(m,k) B[n][k]) with B row-major [N][ldb] (k contiguous; mode 3: [S*S][N][ldb], K =
 *                    KH*KW*OC in total), epilogue 0 none, 1 + bias[n], 2 relu(+ bias[n]), 3 zero where mask[m][n] <= 0 (relu' read
 *                    where the forward left it).  N a multiple of 16, the contraction length (per phase) of 16.
 * pfa_igemm_weights: out (+)= sum_m A(m,k) D[m][n] scattered to torch's parameter layout: perm 0 [k][n], 1 Linear [n][k],
 *                    2 conv [oc=n][ic][ky][kx] from mode-1 k order, 3 the same from mode-2 k order divided by 255, 4 Linear behind an NCHW
 *                    Flatten from NHWC rows (IC, IH, IW of the operand = the flattened tensor); bias_out (nullable) (+)= the column
 *                    sums of D (the bias gradient, from the same pass over D); split over rows, f64 reduction of the splits
 *                    (deterministic).  workspace >= pfa_igemm_weights_workspace_bytes(M, K, N).
 * pfa_colsum:        out[n] (+)= sum_m D[m][n] on its own, f64, deterministic.
 * pfa_cnn_pack_conv / pfa_cnn_transpose: torch weights -> the matrices the loaders' patch orders need (after every optimizer
 *                    step): forward B [OC][k] (u8_order: torch's own order, weights / 255), dX B [S*S][IC][(KH/S)(KW/S)*OC] (nullable);
 *                    Linear [N][K] -> [K][N] (B of its dX).
 * pfa_cnn_heads_sample / pfa_cnn_heads_loss (csrc/cnn_heads.hip): decode_actions + sample_logits, and the PPO loss with its
 *                    gradients w.r.t. the head outputs [rows][16] and the hidden vector [rows][512], for a chunk
 *                    [q0, q0 + rows) of minibatch mb; loss_pairs16 as in pfa_ppo_mlp_grad (accumulate != 0 adds chunks up).
 * pfa_cnn_gather_frames: the frames of such a chunk, contiguous.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t mode;
    int32_t reserved;
    const void *ptr;
    int64_t lda;                                      /* dense only */
    int32_t IC, IH, IW, OC, OH, OW, KH, KW, S;        /* conv geometry (valid padding) */
} pfa_igemm_operand;
int pfa_igemm_rows(const pfa_igemm_operand *a, int64_t M, int32_t K, const float *B, int32_t ldb, int32_t N, float *C, int32_t ldc,
                   int32_t epilogue, const float *bias, const float *mask, int32_t ldmask, pfa_stream_t stream);
/* How the rows form multiplies (process-wide; no reference counterpart — the reference's torch.nn.functional.conv2d / linear
 * (models.py:126-135) leave it to the BLAS): 0 (default) = v_mfma_f32_16x16x4_f32, exact fp32 products in k order; 1 = every fp32
 * operand split into three bf16 pieces and each product issued as its six partial products above 2^-24 on the bf16 matrix path with
 * fp32 accumulation — as close to an f64 product as the fp32 chain, not the same bits.  Contractions that are not a multiple of 32
 * and 16-column outputs keep form 0.  The same switch selects the bf16-path form of the fused 128-wide gradient step
 * (pfa_ppo_mlp_grad_path). */
int pfa_igemm_set_products(int32_t mode);
int pfa_igemm_get_products(void);
size_t pfa_igemm_weights_workspace_bytes(int64_t M, int32_t K, int32_t N);
int pfa_igemm_weights(const pfa_igemm_operand *a, int64_t M, int32_t K, const float *D, int32_t ldd, int32_t N, float *out,
                      int32_t perm, int32_t accumulate, float *bias_out, void *workspace, pfa_stream_t stream);
size_t pfa_colsum_workspace_bytes(int32_t N);
int pfa_colsum(const float *D, int64_t M, int32_t N, int32_t ldd, float *out, int32_t accumulate, void *workspace, pfa_stream_t stream);
int pfa_cnn_pack_conv(const float *w, const pfa_igemm_operand *geom, int32_t u8_order, float *fwd, float *dx, pfa_stream_t stream);
int pfa_cnn_transpose(const float *w, int32_t N, int32_t K, float *out, pfa_stream_t stream);
/* Linear(channels*hw, N) behind nn.Flatten of an NCHW tensor (models.py:133) for NHWC activations: perm_out [N][K'] with the
 * columns in NHWC order (B of the forward), t_out [K'][N] (B of dX); pfa_igemm_weights perm 4 undoes the order for dW. */
int pfa_cnn_pack_fc(const float *w, int32_t N, int32_t channels, int32_t hw, float *perm_out, float *t_out, pfa_stream_t stream);
int pfa_cnn_heads_sample(const float *h, int64_t rows, const float *actor_w, const float *actor_b, const float *value_w,
                         const float *value_b, int32_t num_actions, const float *noise, const pfa_noise_key *key, int64_t row_offset,
                         int64_t *actions, float *logprob, float *entropy, float *value, pfa_stream_t stream);
size_t pfa_cnn_heads_loss_workspace_bytes(void);
int pfa_cnn_heads_loss(const float *h, const pfa_experience *exp, int64_t batch_rows, int32_t mb, int64_t q0, int64_t rows,
                       const float *actor_w, const float *actor_b, const float *value_w, const float *value_b, int32_t num_actions,
                       const pfa_ppo_hparams *hp, const double *adv_stats, int64_t global_mb_rows, float *dout, float *dh,
                       float *loss_pairs16, int32_t accumulate, void *workspace, pfa_stream_t stream);
int pfa_cnn_gather_frames(const uint8_t *frames, int64_t frame_bytes, int64_t batch_rows, int32_t mb, const pfa_ppo_hparams *hp,
                          int64_t q0, int64_t rows, uint8_t *out, pfa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Width-general policy path (csrc/general.hip; host side pufferlib_amd/general.py): pufferlib.models.Default with any
 * hidden_size / observation width, LSTMWrapper with any (input_size, hidden_size) (models.py:24,65) and the recurrent
 * NatureCNN of environments/atari/torch.py:4-6 run as sequences of pfa_igemm_rows / pfa_igemm_weights launches plus these
 * row-wise kernels on the head outputs `out` [rows][ld] (columns < num_actions: the logits of all heads side by side, column
 * num_actions: the value).  num_actions <= 63; heads: the nibble packing of pfa_mlp_dims.heads, 0 = one Discrete head (which may then use all 63).
 *   pfa_heads_rows_sample  sample_logits with action=None (frameworks/cleanrl.py:25-47): actions (packed like pfa_mlp_dims.heads
 *                          for MultiDiscrete), log-prob, entropy (nullable), value
 *   pfa_heads_rows_eval    sample_logits with GIVEN actions (cleanrl.py:38-44): the training-mode policy(obs, action=...) of
 *                          frameworks.cleanrl.Policy / RecurrentPolicy (cleanrl.py:60-66,87-93)
 *   pfa_heads_rows_loss    PPO loss (clean_pufferl.py:202-238) of the chunk's rows and d loss / d out into dout [rows][ldd]
 *                          (columns >= num_out untouched, num_actions < j < num_out zeroed); chunk row r is minibatch row q0 + r, or,
 *                          with time_major_rows = R > 0, row t * R + k is minibatch row q0 + k * bptt_horizon + t;
 *                          loss_pairs16 as in pfa_ppo_mlp_grad (accumulate != 0 adds chunks up)
 *   pfa_lstm_cell_forward  element-wise part of one nn.LSTM step: gates [rows][4 hidden] (i, f, g, o pre-activations, replaced by
 *                          the activated gates), c_prev -> c_out, h_out (row stride ldh) and optionally a second copy h_out2
 *   pfa_lstm_cell_backward its back-propagation: dh = dh_a (+ dh_b), dc in/out (running cell gradient), dgates [rows][4 hidden]
 * ------------------------------------------------------------------------------------------ */
int pfa_heads_rows_sample(const float *out, int32_t ld, int64_t rows, int32_t num_actions, uint32_t heads, const float *noise,
                          const pfa_noise_key *key, int64_t row_offset, int64_t *actions, float *logprob, float *entropy,
                          float *value, pfa_stream_t stream);
int pfa_heads_rows_eval(const float *out, int32_t ld, int64_t rows, int32_t num_actions, uint32_t heads, const int64_t *actions,
                        float *logprob, float *entropy, float *value, pfa_stream_t stream);
size_t pfa_heads_rows_loss_workspace_bytes(int64_t rows);
int pfa_heads_rows_loss(const float *out, int32_t ld, const pfa_experience *exp, int64_t batch_rows, int32_t mb, int64_t q0,
                        int64_t rows, int32_t time_major_rows, int32_t num_actions, uint32_t heads, const pfa_ppo_hparams *hp,
                        const double *adv_stats, int64_t global_mb_rows, float *dout, int32_t ldd, int32_t num_out,
                        float *loss_pairs16, int32_t accumulate, void *workspace, pfa_stream_t stream);
int pfa_lstm_cell_forward(float *gates, const float *c_prev, float *c_out, float *h_out, int32_t ldh, float *h_out2, int32_t ldh2,
                          int64_t rows, int32_t hidden, pfa_stream_t stream);
int pfa_lstm_cell_backward(const float *dh_a, int32_t lda, const float *dh_b, int32_t ldb, float *dc, const float *gates_act,
                           const float *c_prev, const float *c, float *dgates, int64_t rows, int32_t hidden, pfa_stream_t stream);
/* Row copy between the two row orders of a chunk of `segments` x `steps` rows — segment-major (row k * steps + t) and time-major
 * (row t * segments + k) — optionally masked by relu' of `act` (nullable; SRC order, row stride lda): dst = act > 0 ? src : 0.
 * segments == 0: row for row.  cols a multiple of 4, strides multiples of 4, buffers 16-byte aligned. */
int pfa_rows_perm(const float *src, int32_t lds_, float *dst, int32_t ldd, const float *act, int32_t lda, int64_t rows, int32_t cols,
                  int32_t segments, int32_t steps, int32_t to_time_major, pfa_stream_t stream);

/* One-shot all-reduce over peer-mapped device memory (csrc/p2p.hip) for the small buckets of the data-parallel update: every
 * rank writes its bucket into a slot of every peer's buffer over the xGMI mesh, raises a flag, waits for the peers' flags and
 * sums the slots in rank order (one hop instead of a ring's 2 (R-1); identical bits on every rank).  pfa_p2p_alloc creates
 * this rank's fine-grained buffer (slots of cap_bytes) and returns its 64-byte IPC handle; the caller gathers all handles
 * (any transport) and passes them, rank order, to pfa_p2p_open.  Once open, pfa_dist_all_reduce_* and the native train loop
 * use it for every bucket that fits.  World size <= 8 (one node).  The flag waits are bounded (PFA_WAIT_TIMEOUT_MS, default
 * 30 000): a wait that runs out fills its part of the bucket with NaN and raises the status word.  pfa_p2p_status (a plain
 * host read, no synchronisation): 0 ok, 1 a wait of this rank ran out (a peer never arrived), 2 a peer reported one (the ranks' status
 * words ride every flag-in-data exchange: csrc/p2p_ll.hpp), -1 not open.
 * Recovery: pfa_p2p_seq() = this rank's sequence number; after agreeing on base >= every rank's (any transport), every rank calls
 * pfa_p2p_reset(base) between two barriers: clears the status word and restarts both sequence counters from base (nothing a late peer
 * still writes with an old number can satisfy a wait of the new epoch).  The replicas' parameters are
 * the caller's to repair (reload / re-broadcast).  pfa_p2p_debug_set_status: tests only. */
int pfa_p2p_alloc(int64_t cap_bytes, int32_t world, uint8_t *handle64_host);
int pfa_p2p_open(const uint8_t *handles_host, int32_t rank, int32_t world);
int pfa_p2p_close(void);
int pfa_p2p_status(void);
int64_t pfa_p2p_seq(void);
int pfa_p2p_reset(int64_t base);
int pfa_p2p_debug_set_status(int value);
int pfa_p2p_all_reduce_f32(float *buf, int64_t count, pfa_stream_t stream);
int pfa_p2p_all_reduce_f64(double *buf, int64_t count, pfa_stream_t stream);
/* The flag-in-data form of the same exchange (csrc/p2p_ll.hpp): every float travels as one 8-byte {value, sequence number} store
 * into the peers' memory and is summed, rank order, as soon as the local copies carry this call's number — no flag, no fence, no
 * assumption about the order in which a peer's stores land.  This is what pfa_ppo_mlp_train runs INSIDE its reduce + Adam launch
 * when data parallel (no all-reduce launch at all); the stand-alone entry serves the start-up self-test and the tests.  Up to
 * cap_bytes / 4 + 2303 floats (the slot's last entry carries the status words); collective (every rank must make the same sequence of calls).  pfa_p2p_ll_calls: exchanges so far
 * (stand-alone + fused). */
int pfa_p2p_ll_all_reduce_f32(float *buf, int64_t count, pfa_stream_t stream);
int64_t pfa_p2p_ll_calls(void);
/* Peer-wait telemetry: out4 = { ticks, workgroups } of the flag-in-data exchanges (the optimizer steps' exchange inside the reduce +
 * Adam launch), { ticks, chunks } of the flag-based all-reduces (the small f64 exchanges): 100 MHz ticks a workgroup stood waiting
 * for its slowest peer (its longest lane), summed since the last reset.  ticks / workgroups = the mean wait of a launch — transport
 * latency when the ranks arrive together, rank skew on top when they do not.  Synchronises the device; reset != 0 clears. */
int pfa_p2p_wait_stats(int64_t *out4_host, int reset);
/* Routing switch for A/B measurements (call it on every rank alike): on = 0 makes pfa_dist_all_reduce_* and pfa_ppo_mlp_train stop
 * using the peer path — they fall to the RCCL communicator of pfa_dist_init — while the buffers stay mapped; on = 1 routes over it
 * again.  Returns the previous setting. */
int pfa_p2p_enable(int on);

/* What train() logs (clean_pufferl.py:249-254,266-270) in one device buffer of 10 f64: out[0..5] = `losses` (the six
 * running means above), out[6..9] = sum y_true, sum y_true^2, sum adv, sum adv^2 with y_pred = values in storage
 * (step-major) order and y_true = advantages (env-major) + y_pred — the reference's explained-variance inputs. */
int pfa_train_log_sums(const pfa_experience *exp, int64_t batch_rows, int32_t num_envs, const double *losses,
                       double *out10, void *workspace, pfa_stream_t stream);

/* The same in two halves, for the data-parallel update: the four explained-variance sums are known as soon as GAE has run, so
 * they ride the all-reduce of the advantage sums (one exchange instead of two).  pfa_train_ev_sums: out4 = this rank's four sums;
 * pfa_train_log_pack: out10 = { losses[0..5] (nullable: zeros), ev4[0..3] }. */
int pfa_train_ev_sums(const pfa_experience *exp, int64_t batch_rows, int32_t num_envs, double *out4, void *workspace,
                      pfa_stream_t stream);
int pfa_train_log_pack(const double *losses, const double *ev4, double *out10, pfa_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
