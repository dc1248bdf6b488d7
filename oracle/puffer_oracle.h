/* puffer_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the integer / sequential part of the PufferLib 1.0.1 PPO hot path
 * (SURVEY.md §8a rows a2–a8, a16).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link or call this; the product path (pufferlib_amd/) never does.
 *
 * Each function cites the reference file:line it follows (paths relative to /root/reference).
 * Parity status: PINNED — tests/test_oracle_golden.py checks every function here against
 * fixtures in tests/golden/ that were produced by running the unmodified reference in the
 * build container (tests/golden/make_golden.py), plus the known answers in SURVEY.md App. B.
 *
 * Third-party arithmetic restated here (absent from /root/reference):
 *   CPython 3.10.12 stdlib `random` (Lib/random.py: seed, getrandbits, _randbelow_with_getrandbits,
 *   sample; Modules/_randommodule.c: init_by_array seeding) on top of MT19937
 *   (Matsumoto & Nishimura 1998/2002, mt19937ar).  Anchored on the reference's call sites
 *   pufferlib/environments/ocean/ocean.py:449-459 and checked against `random` itself.
 */
#ifndef PUFFER_ORACLE_H
#define PUFFER_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- CPython `random` on MT19937 ------------------------------------------------------ */
typedef struct {
    uint32_t mt[624];
    int32_t idx;    /* next word to hand out, 624 = regenerate first */
    uint64_t count; /* words handed out since the last seeding */
} po_mt_t;

void po_mt_seed(po_mt_t *g, uint64_t seed_abs);          /* random.seed(int) */
uint32_t po_mt_u32(po_mt_t *g);                          /* genrand_uint32 */
uint32_t po_mt_getrandbits(po_mt_t *g, int k);           /* k in 1..32 */
uint32_t po_mt_randbelow(po_mt_t *g, uint32_t n);        /* _randbelow_with_getrandbits */
void po_mt_sample(po_mt_t *g, int n, int k, int *out);   /* random.sample(range(n), k) */

/* ---- Serial vecenv of ocean Squared ---------------------------------------------------- */
typedef struct po_squared_vec po_squared_vec;

/* vector.py:78-95 Serial.__init__ over make_squared (ocean/environment.py:28-31). */
po_squared_vec *po_squared_create(int num_envs, int distance_to_target, int num_targets);
void po_squared_free(po_squared_vec *v);
/* vector.py:112-135 Serial.async_reset: env i is reset with seed+i. */
void po_squared_async_reset(po_squared_vec *v, int64_t seed);
/* vector.py:137-156 Serial.send: done envs reset (unseeded), others step. */
void po_squared_send(po_squared_vec *v, const int64_t *actions);

int po_squared_obs_size(const po_squared_vec *v);         /* (2d+1)^2 */
const float *po_squared_observations(const po_squared_vec *v);  /* [N][obs_size] */
const float *po_squared_rewards(const po_squared_vec *v);       /* [N] */
const uint8_t *po_squared_terminals(const po_squared_vec *v);   /* [N] */
const uint8_t *po_squared_truncations(const po_squared_vec *v); /* [N] */
const uint8_t *po_squared_masks(const po_squared_vec *v);       /* [N] */
/* infos emitted by the last async_reset/send (postprocess.py:18-54): one per finished episode. */
int po_squared_num_infos(const po_squared_vec *v);
const int32_t *po_squared_info_env(const po_squared_vec *v);
const double *po_squared_info_return(const po_squared_vec *v);
const int32_t *po_squared_info_length(const po_squared_vec *v);
const double *po_squared_info_score(const po_squared_vec *v);
/* introspection for tests: target cells (x*g+y) currently remaining / drawn at last reset */
void po_squared_targets(const po_squared_vec *v, int env, int *out_cells /* [num_targets] */);
/* word position of the shared stream (number of 32-bit outputs consumed since the last seeding) */
uint64_t po_squared_stream_pos(const po_squared_vec *v);

/* ---- numpy legacy RandomState on MT19937 (numpy/random/mtrand + _legacy: seed(int) = init_genrand; randint over a
 * 32-bit range = masked rejection on genrand_uint32).  Third-party arithmetic restated; checked against numpy itself. */
void po_mt_seed_numpy(po_mt_t *g, uint32_t seed);          /* np.random.seed(int) */

/* numpy legacy helpers on the same generator: rk_double (53-bit) and legacy_gauss (polar Box-Muller with a cached second
 * value; numpy/random/src/legacy/legacy-distributions.c), randint over [0, n) with masked rejection. */
typedef struct {
    po_mt_t mt;
    int has_gauss;
    double gauss;
} po_np_state;
void po_np_seed(po_np_state *s, uint32_t seed);            /* np.random.seed(int): also drops the cached gaussian */
uint32_t po_np_randint(po_np_state *s, uint32_t n);        /* np.random.randint(0, n), n <= 2**32 */
double po_np_randn(po_np_state *s);                        /* np.random.randn() */

/* ---- Serial vecenv of ocean Bandit ----------------------------------------------------------- */
/* vector.py:78-162 Serial over make_bandit (ocean/environment.py:33-37) = GymnasiumPufferEnv + EpisodeStats over
 * ocean.Bandit (ocean.py:8-63): every reset reseeds numpy's global generator with hard_fixed_seed = 42 and draws
 * solution_idx = randint(0, num_actions); a step returns ((action == solution) + randn() * reward_scale) * reward_scale
 * (noise only if reward_noise != 0) and always terminates. */
/* numpy legacy generator, small-integer path: RandomState.randint(low, high, size, dtype=np.int8) — masked rejection on BYTES
 * of buffered 32-bit words, a fresh buffer per call (_bounded_integers: random_bounded_uint8_fill / buffered_uint8). */
void po_np_randint_i8(po_np_state *s, int low, int high, int count, int8_t *out);
/* np.sum of a contiguous float32 array (numpy's pairwise summation: 8 running sums, then the tail) */
float po_np_sum_f32(const float *a, int n);

/* ---- Serial over ocean.environment.make_spaces (ocean/environment.py:66-69; ocean.py:356-404) ----
 * Observation rows are the emulated 108-byte structs {flat: int8[5] @0, image: f32[5][5] @8} (emulation.py:68-110), actions
 * MultiDiscrete([2, 2]) = (flat, image) (emulation.py:111-121).  reset() draws from numpy's process-global generator WITHOUT
 * seeding it (the seed argument is ignored, ocean.py:380), so the stream is whatever np.random.seed last set:
 * po_spaces_seed_global restates that call (clean_pufferl.seed_everything, clean_pufferl.py:596-600). */
typedef struct po_spaces_vec po_spaces_vec;
po_spaces_vec *po_spaces_create(int num_envs);
void po_spaces_free(po_spaces_vec *v);
void po_spaces_seed_global(po_spaces_vec *v, uint32_t seed);
void po_spaces_async_reset(po_spaces_vec *v);
void po_spaces_send(po_spaces_vec *v, const int64_t *actions /* [N][2] */);
const uint8_t *po_spaces_observations(const po_spaces_vec *v); /* [N][108] */
const float *po_spaces_rewards(const po_spaces_vec *v);
const uint8_t *po_spaces_terminals(const po_spaces_vec *v);
int po_spaces_num_infos(const po_spaces_vec *v);
const double *po_spaces_info_score(const po_spaces_vec *v);

typedef struct po_bandit_vec po_bandit_vec;
po_bandit_vec *po_bandit_create(int num_envs, int num_actions, double reward_scale, double reward_noise);
void po_bandit_free(po_bandit_vec *v);
void po_bandit_async_reset(po_bandit_vec *v, int64_t seed);
void po_bandit_send(po_bandit_vec *v, const int64_t *actions);
const float *po_bandit_observations(const po_bandit_vec *v); /* [N][1] */
const float *po_bandit_rewards(const po_bandit_vec *v);
const uint8_t *po_bandit_terminals(const po_bandit_vec *v);
const uint8_t *po_bandit_masks(const po_bandit_vec *v);
int po_bandit_num_infos(const po_bandit_vec *v);
const double *po_bandit_info_return(const po_bandit_vec *v);
const double *po_bandit_info_score(const po_bandit_vec *v);
int po_bandit_solution(const po_bandit_vec *v);

/* ---- Serial vecenv of ocean Memory ---------------------------------------------------------- */
/* vector.py:78-162 Serial over make_memory (ocean/environment.py:41-44) = GymnasiumPufferEnv + EpisodeStats over
 * ocean.Memory (ocean.py:65-123).  reset(seed) seeds numpy's PROCESS-GLOBAL generator when a seed is given (async_reset:
 * seed + i per env) and always draws solution = np.random.randint(0, 2, size=horizon) from it, so after async_reset every
 * env shares the stream left by the last env. */
typedef struct po_memory_vec po_memory_vec;
po_memory_vec *po_memory_create(int num_envs, int mem_length, int mem_delay);
void po_memory_free(po_memory_vec *v);
void po_memory_async_reset(po_memory_vec *v, int64_t seed);
void po_memory_send(po_memory_vec *v, const int64_t *actions);
const float *po_memory_observations(const po_memory_vec *v); /* [N][1] */
const float *po_memory_rewards(const po_memory_vec *v);
const uint8_t *po_memory_terminals(const po_memory_vec *v);
const uint8_t *po_memory_truncations(const po_memory_vec *v);
const uint8_t *po_memory_masks(const po_memory_vec *v);
int po_memory_num_infos(const po_memory_vec *v);
const int32_t *po_memory_info_env(const po_memory_vec *v);
const double *po_memory_info_return(const po_memory_vec *v);
const int32_t *po_memory_info_length(const po_memory_vec *v);
const double *po_memory_info_score(const po_memory_vec *v);
void po_memory_solution(const po_memory_vec *v, int env, float *out /* [horizon] */);

/* ---- Serial vecenv of ocean Stochastic --------------------------------------------------- */
/* vector.py:78-162 Serial over make_stochastic (ocean/environment.py:61-64: horizon fixed to 100) = GymnasiumPufferEnv
 * (emulation.py:169-228) over EpisodeStats (postprocess.py:18-54) over ocean.Stochastic (ocean.py:529-582).  The env has no
 * randomness: reward = 1 - (p - count/tick)**2 (python float arithmetic, cast to f32 by the buffer write) when the action
 * moves the running action-0 fraction towards p, else 0; observation is always [0.0]. */
typedef struct po_stochastic_vec po_stochastic_vec;
po_stochastic_vec *po_stochastic_create(int num_envs, double p, int horizon);
void po_stochastic_free(po_stochastic_vec *v);
void po_stochastic_async_reset(po_stochastic_vec *v, int64_t seed);
void po_stochastic_send(po_stochastic_vec *v, const int64_t *actions);
const float *po_stochastic_observations(const po_stochastic_vec *v); /* [N][1] */
const float *po_stochastic_rewards(const po_stochastic_vec *v);
const uint8_t *po_stochastic_terminals(const po_stochastic_vec *v);
const uint8_t *po_stochastic_truncations(const po_stochastic_vec *v);
const uint8_t *po_stochastic_masks(const po_stochastic_vec *v);
int po_stochastic_num_infos(const po_stochastic_vec *v);
const int32_t *po_stochastic_info_env(const po_stochastic_vec *v);
const double *po_stochastic_info_return(const po_stochastic_vec *v);
const int32_t *po_stochastic_info_length(const po_stochastic_vec *v);
const double *po_stochastic_info_score(const po_stochastic_vec *v);
/* the reward ocean.Stochastic.step returns after `tick` steps of which `count` were action 0, the last one `action` */
double po_stochastic_reward(double p, int tick, int count, int action, double *proximity_out);

/* ---- GAE ---------------------------------------------------------------------------------- */
/* c_gae.pyx:11-32 */
void po_compute_gae(const float *dones, const float *values, const float *rewards,
                    float *advantages, int num_steps, float gamma, float gae_lambda);

/* ---- counter-based action noise (our own definition, restated for checking) --------------- */
/* Philox4x32-10 keyed by (seed_lo, seed_hi); counter = (row, j, step_lo, step_hi); the four output
 * words give exponential(1) variates q = -log(u), u = ((w >> 8) + 0.5) * 2^-24, for action columns
 * 4*j .. 4*j+3.  See pufferlib_amd/csrc/philox.hpp. */
void po_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
