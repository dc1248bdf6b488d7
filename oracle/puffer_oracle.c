/* puffer_oracle.c — CPU ORACLE (test infrastructure, NOT product code).  See puffer_oracle.h.
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off oracle/puffer_oracle.c -o oracle/_build/libpuffer_oracle.so
 * (-ffp-contract=off: the reference's Cython GAE is compiled without FMA contraction on x86-64.)
 */
#include "puffer_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================================== */
/* MT19937 + CPython `random`                                                                */
/* ======================================================================================== */
#define MT_N 624
#define MT_M 397

static void mt_init_genrand(po_mt_t *g, uint32_t s) {
    g->mt[0] = s;
    for (int i = 1; i < MT_N; i++)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = MT_N;
    g->count = 0;
}

static void mt_init_by_array(po_mt_t *g, const uint32_t *key, int key_len) {
    mt_init_genrand(g, 19650218u);
    uint32_t *mt = g->mt;
    int i = 1, j = 0;
    int k = MT_N > key_len ? MT_N : key_len;
    for (; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++;
        j++;
        if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
        if (j >= key_len) j = 0;
    }
    for (k = MT_N - 1; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
    }
    mt[0] = 0x80000000u;
    g->idx = MT_N;
}

/* random.seed(a) for int a: key = 32-bit little-endian digits of abs(a) (at least one word).
 * Follows CPython Modules/_randommodule.c random_seed(); called from ocean.py:449-450. */
void po_mt_seed(po_mt_t *g, uint64_t seed_abs) {
    uint32_t key[2] = {(uint32_t)(seed_abs & 0xffffffffu), (uint32_t)(seed_abs >> 32)};
    mt_init_by_array(g, key, key[1] ? 2 : 1);
}

static void mt_regenerate(po_mt_t *g) {
    uint32_t *mt = g->mt;
    for (int kk = 0; kk < MT_N; kk++) {
        uint32_t y = (mt[kk] & 0x80000000u) | (mt[(kk + 1) % MT_N] & 0x7fffffffu);
        mt[kk] = mt[(kk + MT_M) % MT_N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->idx = 0;
}

uint32_t po_mt_u32(po_mt_t *g) {
    if (g->idx >= MT_N) mt_regenerate(g);
    uint32_t y = g->mt[g->idx++];
    g->count++;
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* Random.getrandbits(k), k <= 32: top k bits of one output word. */
uint32_t po_mt_getrandbits(po_mt_t *g, int k) { return po_mt_u32(g) >> (32 - k); }

static int bit_length(uint32_t n) {
    int k = 0;
    while (n) { k++; n >>= 1; }
    return k;
}

/* Lib/random.py Random._randbelow_with_getrandbits */
uint32_t po_mt_randbelow(po_mt_t *g, uint32_t n) {
    if (!n) return 0;
    int k = bit_length(n);
    uint32_t r = po_mt_getrandbits(g, k);
    while (r >= n) r = po_mt_getrandbits(g, k);
    return r;
}

/* Lib/random.py Random.sample(population, k) restricted to index output:
 * returns the k chosen positions of a length-n population. */
void po_mt_sample(po_mt_t *g, int n, int k, int *out) {
    int setsize = 21;
    if (k > 5) { /* 4 ** ceil(log(3k, 4)) */
        int p = 1;
        while (p < 3 * k) p *= 4;
        setsize += p;
    }
    if (n <= setsize) {
        int *pool = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
        for (int i = 0; i < n; i++) pool[i] = i;
        for (int i = 0; i < k; i++) {
            int j = (int)po_mt_randbelow(g, (uint32_t)(n - i));
            out[i] = pool[j];
            pool[j] = pool[n - i - 1];
        }
        free(pool);
    } else {
        for (int i = 0; i < k; i++) {
            int j;
            for (;;) {
                j = (int)po_mt_randbelow(g, (uint32_t)n);
                int dup = 0;
                for (int q = 0; q < i; q++) dup |= (out[q] == j);
                if (!dup) break;
            }
            out[i] = j;
        }
    }
}

/* ======================================================================================== */
/* Serial vecenv of ocean Squared                                                            */
/* ======================================================================================== */
static const int MOVES[8][2] = {/* ocean.py:424 */
                                {0, -1}, {0, 1}, {-1, 0}, {1, 0}, {1, -1}, {-1, -1}, {1, 1}, {-1, 1}};

typedef struct {
    int x, y, tick;
    int n_remaining;
    int *targets; /* remaining target cells (x*g+y), insertion order (list.remove semantics) */
    int done;     /* GymnasiumPufferEnv.done, emulation.py:130,226 */
    /* EpisodeStats (postprocess.py:18-31) */
    double ret_sum;
    int ep_len;
} sq_env;

struct po_squared_vec {
    int n, d, nt, g, obs_size, n_perim;
    int *perim; /* possible_targets, row-major over perimeter cells (ocean.py:444-446) */
    sq_env *envs;
    int *target_store;
    float *obs;
    float *rewards;
    uint8_t *terminals, *truncations, *masks;
    po_mt_t rng; /* the process-global `random` generator */
    int n_infos;
    int32_t *info_env, *info_len;
    double *info_ret, *info_score;
    int *scratch;
};

static int iabs(int a) { return a < 0 ? -a : a; }
static int imax(int a, int b) { return a > b ? a : b; }

po_squared_vec *po_squared_create(int num_envs, int d, int nt) {
    po_squared_vec *v = (po_squared_vec *)calloc(1, sizeof(*v));
    v->n = num_envs;
    v->d = d;
    v->g = 2 * d + 1;
    if (nt == -1) nt = 4 * d; /* ocean.py:432-433 */
    v->nt = nt;
    v->obs_size = v->g * v->g;
    v->perim = (int *)malloc(sizeof(int) * (size_t)(8 * d + 1));
    v->n_perim = 0;
    for (int x = 0; x < v->g; x++)
        for (int y = 0; y < v->g; y++)
            if (x == 0 || y == 0 || x == v->g - 1 || y == v->g - 1) v->perim[v->n_perim++] = x * v->g + y;
    v->envs = (sq_env *)calloc((size_t)num_envs, sizeof(sq_env));
    v->target_store = (int *)calloc((size_t)num_envs * (size_t)(nt > 0 ? nt : 1), sizeof(int));
    for (int i = 0; i < num_envs; i++) {
        v->envs[i].targets = v->target_store + (size_t)i * (size_t)nt;
        v->envs[i].done = 1;
    }
    v->obs = (float *)calloc((size_t)num_envs * (size_t)v->obs_size, sizeof(float));
    v->rewards = (float *)calloc((size_t)num_envs, sizeof(float));
    v->terminals = (uint8_t *)calloc((size_t)num_envs, 1);
    v->truncations = (uint8_t *)calloc((size_t)num_envs, 1);
    v->masks = (uint8_t *)malloc((size_t)num_envs);
    memset(v->masks, 1, (size_t)num_envs); /* vector.py:124 */
    v->info_env = (int32_t *)calloc((size_t)num_envs, sizeof(int32_t));
    v->info_len = (int32_t *)calloc((size_t)num_envs, sizeof(int32_t));
    v->info_ret = (double *)calloc((size_t)num_envs, sizeof(double));
    v->info_score = (double *)calloc((size_t)num_envs, sizeof(double));
    v->scratch = (int *)calloc((size_t)(nt > 0 ? nt : 1), sizeof(int));
    po_mt_seed(&v->rng, 0);
    return v;
}

void po_squared_free(po_squared_vec *v) {
    if (!v) return;
    free(v->perim); free(v->envs); free(v->target_store); free(v->obs); free(v->rewards);
    free(v->terminals); free(v->truncations); free(v->masks); free(v->info_env); free(v->info_len);
    free(v->info_ret); free(v->info_score); free(v->scratch); free(v);
}

/* ocean.py:448-463 Squared.reset + emulation.py:169-192 GymnasiumPufferEnv.reset +
 * postprocess.py:18-20 EpisodeStats.reset. */
static void env_reset(po_squared_vec *v, int i, int has_seed, int64_t seed) {
    sq_env *e = &v->envs[i];
    if (has_seed) {
        po_mt_seed(&v->rng, (uint64_t)(seed < 0 ? -seed : seed));
    }
    float *grid = v->obs + (size_t)i * (size_t)v->obs_size;
    memset(grid, 0, sizeof(float) * (size_t)v->obs_size);
    grid[v->d * v->g + v->d] = -1.0f;
    e->x = v->d;
    e->y = v->d;
    e->tick = 0;
    po_mt_sample(&v->rng, v->n_perim, v->nt, v->scratch); /* random.sample(possible_targets, nt) */
    e->n_remaining = v->nt;
    for (int t = 0; t < v->nt; t++) {
        int cell = v->perim[v->scratch[t]];
        e->targets[t] = cell;
        grid[cell] = 1.0f;
    }
    e->done = 0;
    e->ret_sum = 0.0;
    e->ep_len = 0;
    v->rewards[i] = 0.0f;
    v->terminals[i] = 0;
    v->truncations[i] = 0;
    v->masks[i] = 1;
}

/* ocean.py:465-513 Squared.step + postprocess.py:22-54 EpisodeStats.step +
 * emulation.py:194-228 GymnasiumPufferEnv.step. */
static void env_step(po_squared_vec *v, int i, int64_t action) {
    sq_env *e = &v->envs[i];
    const int g = v->g, d = v->d;
    float *grid = v->obs + (size_t)i * (size_t)v->obs_size;
    int x = e->x, y = e->y;
    grid[x * g + y] = 0.0f;
    x += MOVES[action][0];
    y += MOVES[action][1];
    int min_dist = 1 << 30;
    for (int t = 0; t < e->n_remaining; t++) {
        int tx = e->targets[t] / g, ty = e->targets[t] % g;
        int dist = imax(iabs(x - tx), iabs(y - ty));
        if (dist < min_dist) min_dist = dist;
    }
    double reward = 1.0 - (double)min_dist / (double)d; /* python float arithmetic */
    for (int t = 0; t < e->n_remaining; t++) {
        if (e->targets[t] == x * g + y) { /* list.remove: first match, keep order */
            for (int u = t; u + 1 < e->n_remaining; u++) e->targets[u] = e->targets[u + 1];
            e->n_remaining--;
            break;
        }
    }
    int dist_from_origin = imax(iabs(x - d), iabs(y - d));
    if (dist_from_origin >= d) { x = d; y = d; }
    e->x = x;
    e->y = y;
    grid[x * g + y] = -1.0f;
    e->tick += 1;
    int done = e->tick >= v->nt * d; /* max_ticks, ocean.py:438 */

    e->ret_sum += reward; /* sum(list of python floats), left to right from 0 */
    e->ep_len += 1;
    if (done) {
        int k = v->n_infos++;
        v->info_env[k] = i;
        v->info_ret[k] = e->ret_sum;
        v->info_len[k] = e->ep_len;
        v->info_score[k] = (double)(v->nt - e->n_remaining) / (double)v->nt;
    }
    v->rewards[i] = (float)reward; /* emulation.py:221, double -> float32 */
    v->terminals[i] = (uint8_t)done;
    v->truncations[i] = 0;
    v->masks[i] = 1;
    e->done = done;
}

void po_squared_async_reset(po_squared_vec *v, int64_t seed) {
    v->n_infos = 0;
    for (int i = 0; i < v->n; i++) env_reset(v, i, 1, seed + i); /* vector.py:114,129-130,639-641 */
}

void po_squared_send(po_squared_vec *v, const int64_t *actions) {
    v->n_infos = 0;
    for (int i = 0; i < v->n; i++) {
        if (v->envs[i].done)
            env_reset(v, i, 0, 0); /* vector.py:147-149: action ignored, unseeded reset */
        else
            env_step(v, i, actions[i]);
    }
}

int po_squared_obs_size(const po_squared_vec *v) { return v->obs_size; }
const float *po_squared_observations(const po_squared_vec *v) { return v->obs; }
const float *po_squared_rewards(const po_squared_vec *v) { return v->rewards; }
const uint8_t *po_squared_terminals(const po_squared_vec *v) { return v->terminals; }
const uint8_t *po_squared_truncations(const po_squared_vec *v) { return v->truncations; }
const uint8_t *po_squared_masks(const po_squared_vec *v) { return v->masks; }
int po_squared_num_infos(const po_squared_vec *v) { return v->n_infos; }
const int32_t *po_squared_info_env(const po_squared_vec *v) { return v->info_env; }
const double *po_squared_info_return(const po_squared_vec *v) { return v->info_ret; }
const int32_t *po_squared_info_length(const po_squared_vec *v) { return v->info_len; }
const double *po_squared_info_score(const po_squared_vec *v) { return v->info_score; }
void po_squared_targets(const po_squared_vec *v, int env, int *out_cells) {
    for (int t = 0; t < v->nt; t++) out_cells[t] = t < v->envs[env].n_remaining ? v->envs[env].targets[t] : -1;
}
uint64_t po_squared_stream_pos(const po_squared_vec *v) { return v->rng.count; }

/* ======================================================================================== */
/* numpy legacy seeding + Serial(make_memory) — ocean.py:65-123                                 */
/* ======================================================================================== */
void po_mt_seed_numpy(po_mt_t *g, uint32_t seed) { /* numpy/random/src/mt19937/mt19937.c: mt19937_seed = init_genrand */
    g->mt[0] = seed;
    for (int i = 1; i < 624; i++) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
    g->count = 0;
}

void po_np_seed(po_np_state *s, uint32_t seed) {
    po_mt_seed_numpy(&s->mt, seed);
    s->has_gauss = 0; /* _legacy_seeding clears the cached gaussian */
    s->gauss = 0.0;
}

uint32_t po_np_randint(po_np_state *s, uint32_t n) { /* _bounded_integers: masked rejection, range n - 1 */
    const uint32_t rng = n - 1;
    if (rng == 0) return 0;
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    while ((v = po_mt_u32(&s->mt) & mask) > rng) {}
    return v;
}

static double np_double(po_np_state *s) { /* mt19937_next_double */
    const uint32_t a = po_mt_u32(&s->mt) >> 5, b = po_mt_u32(&s->mt) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

double po_np_randn(po_np_state *s) { /* legacy_gauss */
    if (s->has_gauss) {
        const double t = s->gauss;
        s->has_gauss = 0;
        s->gauss = 0.0;
        return t;
    }
    double f, x1, x2, r2;
    do {
        x1 = 2.0 * np_double(s) - 1.0;
        x2 = 2.0 * np_double(s) - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = sqrt(-2.0 * log(r2) / r2);
    s->gauss = f * x1;
    s->has_gauss = 1;
    return f * x2;
}

void po_np_randint_i8(po_np_state *s, int low, int high, int count, int8_t *out) {
    const uint8_t rng = (uint8_t)(high - 1 - low), off = (uint8_t)low; /* closed range, offset wraps like the C cast */
    uint32_t buf = 0;
    int bcnt = 0;
    uint8_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4;
    for (int i = 0; i < count; i++) {
        if (rng == 0) {
            out[i] = (int8_t)off;
            continue;
        }
        uint8_t val;
        do { /* buffered_bounded_masked_uint8 */
            if (!bcnt) {
                buf = po_mt_u32(&s->mt);
                bcnt = 3;
            } else {
                buf >>= 8;
                bcnt -= 1;
            }
            val = (uint8_t)buf & mask;
        } while (val > rng);
        out[i] = (int8_t)(uint8_t)(off + val);
    }
}

float po_np_sum_f32(const float *a, int n) { /* numpy/core/src/umath/loops_utils.h.src: pairwise sum, blocks below 128 elements */
    if (n < 8) {
        float res = 0.0f;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    float r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

/* ---- Serial(make_spaces) — ocean.py:356-404 ---- */
#define PO_SPACES_ROW 108
struct po_spaces_vec {
    int n, n_infos;
    po_np_state rng; /* np.random's global state */
    int *done, *image_sign, *flat_sign;
    uint8_t *obs, *terminals;
    float *rewards;
    double *info_score;
};
po_spaces_vec *po_spaces_create(int num_envs) {
    po_spaces_vec *v = (po_spaces_vec *)calloc(1, sizeof(*v));
    v->n = num_envs;
    v->done = (int *)calloc((size_t)num_envs, sizeof(int));
    v->image_sign = (int *)calloc((size_t)num_envs, sizeof(int));
    v->flat_sign = (int *)calloc((size_t)num_envs, sizeof(int));
    v->obs = (uint8_t *)calloc((size_t)num_envs, PO_SPACES_ROW);
    v->terminals = (uint8_t *)calloc((size_t)num_envs, 1);
    v->rewards = (float *)calloc((size_t)num_envs, sizeof(float));
    v->info_score = (double *)calloc((size_t)num_envs, sizeof(double));
    po_np_seed(&v->rng, 0);
    return v;
}
void po_spaces_free(po_spaces_vec *v) {
    if (!v) return;
    free(v->done); free(v->image_sign); free(v->flat_sign); free(v->obs); free(v->terminals); free(v->rewards); free(v->info_score);
    free(v);
}
void po_spaces_seed_global(po_spaces_vec *v, uint32_t seed) { po_np_seed(&v->rng, seed); }
static void spaces_reset(po_spaces_vec *v, int e) { /* ocean.py:380-389; row layout emulation.py:68-80 (align=True) */
    float image[25];
    int8_t flat[5];
    for (int i = 0; i < 25; i++) image[i] = (float)po_np_randn(&v->rng); /* randn(5, 5).astype(float32) */
    po_np_randint_i8(&v->rng, -1, 2, 5, flat);
    v->image_sign[e] = po_np_sum_f32(image, 25) > 0.0f;
    int fs = 0;
    for (int i = 0; i < 5; i++) fs += flat[i];
    v->flat_sign[e] = fs > 0;
    uint8_t *row = v->obs + (size_t)e * PO_SPACES_ROW;
    memset(row, 0, PO_SPACES_ROW);
    memcpy(row, flat, 5);
    memcpy(row + 8, image, sizeof(image));
    v->done[e] = 0;
    v->rewards[e] = 0.0f;
    v->terminals[e] = 0;
}
void po_spaces_async_reset(po_spaces_vec *v) {
    v->n_infos = 0;
    for (int e = 0; e < v->n; e++) spaces_reset(v, e);
}
void po_spaces_send(po_spaces_vec *v, const int64_t *actions) { /* vector.py:137-156 over ocean.py:391-404 */
    v->n_infos = 0;
    for (int e = 0; e < v->n; e++) {
        if (v->done[e]) {
            spaces_reset(v, e);
            continue;
        }
        double reward = 0.0;
        if (v->image_sign[e] == (int)actions[2 * e + 1]) reward += 0.5; /* Dict keys sorted: action 0 = flat, 1 = image */
        if (v->flat_sign[e] == (int)actions[2 * e]) reward += 0.5;
        v->rewards[e] = (float)reward;
        v->terminals[e] = 1;
        v->done[e] = 1;
        v->info_score[v->n_infos++] = reward;
    }
}
const uint8_t *po_spaces_observations(const po_spaces_vec *v) { return v->obs; }
const float *po_spaces_rewards(const po_spaces_vec *v) { return v->rewards; }
const uint8_t *po_spaces_terminals(const po_spaces_vec *v) { return v->terminals; }
int po_spaces_num_infos(const po_spaces_vec *v) { return v->n_infos; }
const double *po_spaces_info_score(const po_spaces_vec *v) { return v->info_score; }

/* ---- Serial(make_bandit) — ocean.py:8-63 ---- */
struct po_bandit_vec {
    int n, num_actions, n_infos, solution;
    double scale, noise;
    po_np_state rng; /* np.random's global state */
    int *done;
    float *obs, *rewards;
    uint8_t *terminals, *masks;
    double *info_ret, *info_score;
};

po_bandit_vec *po_bandit_create(int num_envs, int num_actions, double reward_scale, double reward_noise) {
    po_bandit_vec *v = (po_bandit_vec *)calloc(1, sizeof(*v));
    v->n = num_envs;
    v->num_actions = num_actions;
    v->scale = reward_scale;
    v->noise = reward_noise;
    v->done = (int *)calloc((size_t)num_envs, sizeof(int));
    v->obs = (float *)calloc((size_t)num_envs, sizeof(float));
    v->rewards = (float *)calloc((size_t)num_envs, sizeof(float));
    v->terminals = (uint8_t *)calloc((size_t)num_envs, 1);
    v->masks = (uint8_t *)malloc((size_t)num_envs);
    memset(v->masks, 1, (size_t)num_envs);
    v->info_ret = (double *)calloc((size_t)num_envs, sizeof(double));
    v->info_score = (double *)calloc((size_t)num_envs, sizeof(double));
    po_np_seed(&v->rng, 0);
    return v;
}
void po_bandit_free(po_bandit_vec *v) {
    if (!v) return;
    free(v->done); free(v->obs); free(v->rewards); free(v->terminals); free(v->masks); free(v->info_ret); free(v->info_score);
    free(v);
}
static void bandit_reset(po_bandit_vec *v, int i) { /* ocean.py:33-44: the seed argument is replaced by hard_fixed_seed = 42 */
    po_np_seed(&v->rng, 42);
    v->solution = (int)po_np_randint(&v->rng, (uint32_t)v->num_actions);
    v->done[i] = 0;
    v->obs[i] = 1.0f;
    v->rewards[i] = 0.0f;
    v->terminals[i] = 0;
    v->masks[i] = 1;
}
void po_bandit_async_reset(po_bandit_vec *v, int64_t seed) {
    (void)seed;
    v->n_infos = 0;
    for (int i = 0; i < v->n; i++) bandit_reset(v, i);
}
void po_bandit_send(po_bandit_vec *v, const int64_t *actions) { /* ocean.py:46-63 */
    v->n_infos = 0;
    for (int i = 0; i < v->n; i++) {
        if (v->done[i]) {
            bandit_reset(v, i);
            continue;
        }
        const int correct = (int)actions[i] == v->solution;
        double noise = 0.0;
        if (v->noise != 0.0) noise = po_np_randn(&v->rng) * v->scale;
        const double reward = ((double)correct + noise) * v->scale;
        v->obs[i] = 1.0f;
        v->rewards[i] = (float)reward;
        v->terminals[i] = 1;
        v->masks[i] = 1;
        v->done[i] = 1;
        const int k = v->n_infos++;
        v->info_ret[k] = reward;      /* sum([reward]) */
        v->info_score[k] = (double)correct;
    }
}
const float *po_bandit_observations(const po_bandit_vec *v) { return v->obs; }
const float *po_bandit_rewards(const po_bandit_vec *v) { return v->rewards; }
const uint8_t *po_bandit_terminals(const po_bandit_vec *v) { return v->terminals; }
const uint8_t *po_bandit_masks(const po_bandit_vec *v) { return v->masks; }
int po_bandit_num_infos(const po_bandit_vec *v) { return v->n_infos; }
const double *po_bandit_info_return(const po_bandit_vec *v) { return v->info_ret; }
const double *po_bandit_info_score(const po_bandit_vec *v) { return v->info_score; }
int po_bandit_solution(const po_bandit_vec *v) { return v->solution; }

#define PO_MEM_MAXH 64
typedef struct {
    int tick, done, ep_length;
    float solution[PO_MEM_MAXH], submission[PO_MEM_MAXH];
    double ep_return;
} mem_env;

struct po_memory_vec {
    int n, L, D, H, n_infos;
    po_mt_t rng; /* np.random's global state */
    mem_env *envs;
    float *obs, *rewards;
    uint8_t *terminals, *truncations, *masks;
    int32_t *info_env, *info_len;
    double *info_ret, *info_score;
};

po_memory_vec *po_memory_create(int num_envs, int mem_length, int mem_delay) {
    po_memory_vec *v = (po_memory_vec *)calloc(1, sizeof(*v));
    v->n = num_envs;
    v->L = mem_length;
    v->D = mem_delay;
    v->H = 2 * mem_length + mem_delay; /* ocean.py:83 */
    if (v->H > PO_MEM_MAXH || v->H < 2) { free(v); return NULL; }
    v->envs = (mem_env *)calloc((size_t)num_envs, sizeof(mem_env));
    v->obs = (float *)calloc((size_t)num_envs, sizeof(float));
    v->rewards = (float *)calloc((size_t)num_envs, sizeof(float));
    v->terminals = (uint8_t *)calloc((size_t)num_envs, 1);
    v->truncations = (uint8_t *)calloc((size_t)num_envs, 1);
    v->masks = (uint8_t *)malloc((size_t)num_envs);
    memset(v->masks, 1, (size_t)num_envs);
    v->info_env = (int32_t *)calloc((size_t)num_envs, sizeof(int32_t));
    v->info_len = (int32_t *)calloc((size_t)num_envs, sizeof(int32_t));
    v->info_ret = (double *)calloc((size_t)num_envs, sizeof(double));
    v->info_score = (double *)calloc((size_t)num_envs, sizeof(double));
    for (int i = 0; i < num_envs; i++) v->envs[i].done = 1;
    po_mt_seed_numpy(&v->rng, 0);
    return v;
}

void po_memory_free(po_memory_vec *v) {
    if (!v) return;
    free(v->envs); free(v->obs); free(v->rewards); free(v->terminals); free(v->truncations); free(v->masks);
    free(v->info_env); free(v->info_len); free(v->info_ret); free(v->info_score); free(v);
}

/* ocean.py:90-100 reset + emulation.py:169-192 + postprocess.py:18-20 */
static void mem_reset(po_memory_vec *v, int i, int has_seed, int64_t seed) {
    mem_env *e = &v->envs[i];
    if (has_seed) po_mt_seed_numpy(&v->rng, (uint32_t)seed); /* np.random.seed(seed), ocean.py:92-93 */
    /* np.random.randint(0, 2, size=horizon): range 1 -> mask 1, one 32-bit word per element, never rejected */
    for (int j = 0; j < v->H; j++) e->solution[j] = (float)(po_mt_u32(&v->rng) & 1u);
    for (int j = v->H - (v->L + v->D); j < v->H; j++) e->solution[j] = -1.0f; /* ocean.py:96 */
    for (int j = 0; j < v->H; j++) e->submission[j] = -1.0f;
    e->tick = 1;
    e->done = 0;
    e->ep_return = 0.0;
    e->ep_length = 0;
    v->obs[i] = e->solution[0];
    v->rewards[i] = 0.0f;
    v->terminals[i] = 0;
    v->truncations[i] = 0;
    v->masks[i] = 1;
}

void po_memory_async_reset(po_memory_vec *v, int64_t seed) {
    v->n_infos = 0;
    for (int i = 0; i < v->n; i++) mem_reset(v, i, 1, seed + i); /* vector.py:129-130 */
}

/* ocean.py:102-123 */
void po_memory_send(po_memory_vec *v, const int64_t *actions) {
    v->n_infos = 0;
    for (int i = 0; i < v->n; i++) {
        mem_env *e = &v->envs[i];
        if (e->done) { /* vector.py:147-149: action ignored, unseeded reset draws from the shared stream */
            mem_reset(v, i, 0, 0);
            continue;
        }
        const int a = (int)actions[i];
        float ob = 0.0f;
        double reward = 0.0;
        if (e->tick < v->L) {
            ob = e->solution[e->tick];
            reward = a == 0 ? 1.0 : 0.0;
        }
        if (e->tick >= v->L + v->D) {
            const int idx = e->tick - v->L - v->D;
            reward = (float)a == e->solution[idx] ? 1.0 : 0.0;
            e->submission[e->tick] = (float)a;
        }
        e->tick += 1;
        const int terminal = e->tick == v->H;
        e->ep_return += reward;
        e->ep_length += 1;
        v->obs[i] = ob;
        v->rewards[i] = (float)reward;
        v->terminals[i] = (uint8_t)terminal;
        v->truncations[i] = 0;
        v->masks[i] = 1;
        e->done = terminal;
        if (terminal) {
            int ok = 1; /* np.all(solution[:L] == submission[-L:]), ocean.py:120-121 */
            for (int j = 0; j < v->L; j++) ok &= e->solution[j] == e->submission[v->H - v->L + j];
            const int k = v->n_infos++;
            v->info_env[k] = i;
            v->info_ret[k] = e->ep_return;
            v->info_len[k] = e->ep_length;
            v->info_score[k] = (double)ok;
        }
    }
}

const float *po_memory_observations(const po_memory_vec *v) { return v->obs; }
const float *po_memory_rewards(const po_memory_vec *v) { return v->rewards; }
const uint8_t *po_memory_terminals(const po_memory_vec *v) { return v->terminals; }
const uint8_t *po_memory_truncations(const po_memory_vec *v) { return v->truncations; }
const uint8_t *po_memory_masks(const po_memory_vec *v) { return v->masks; }
int po_memory_num_infos(const po_memory_vec *v) { return v->n_infos; }
const int32_t *po_memory_info_env(const po_memory_vec *v) { return v->info_env; }
const double *po_memory_info_return(const po_memory_vec *v) { return v->info_ret; }
const int32_t *po_memory_info_length(const po_memory_vec *v) { return v->info_len; }
const double *po_memory_info_score(const po_memory_vec *v) { return v->info_score; }
void po_memory_solution(const po_memory_vec *v, int env, float *out) {
    for (int j = 0; j < v->H; j++) out[j] = v->envs[env].solution[j];
}

/* ======================================================================================== */
/* Serial(make_stochastic) — ocean.py:529-582, emulation.py:169-228, postprocess.py:18-54        */
/* ======================================================================================== */
typedef struct {
    int tick, count, done;
    double ep_return; /* python sum() of the episode's rewards, in order */
    int ep_length;
} st_env;

struct po_stochastic_vec {
    int n, horizon, n_infos;
    double p;
    st_env *envs;
    float *obs, *rewards;
    uint8_t *terminals, *truncations, *masks;
    int32_t *info_env, *info_len;
    double *info_ret, *info_score;
};

po_stochastic_vec *po_stochastic_create(int num_envs, double p, int horizon) {
    po_stochastic_vec *v = (po_stochastic_vec *)calloc(1, sizeof(*v));
    v->n = num_envs;
    v->p = p;
    v->horizon = horizon;
    v->envs = (st_env *)calloc((size_t)num_envs, sizeof(st_env));
    v->obs = (float *)calloc((size_t)num_envs, sizeof(float));
    v->rewards = (float *)calloc((size_t)num_envs, sizeof(float));
    v->terminals = (uint8_t *)calloc((size_t)num_envs, 1);
    v->truncations = (uint8_t *)calloc((size_t)num_envs, 1);
    v->masks = (uint8_t *)malloc((size_t)num_envs);
    memset(v->masks, 1, (size_t)num_envs);
    v->info_env = (int32_t *)calloc((size_t)num_envs, sizeof(int32_t));
    v->info_len = (int32_t *)calloc((size_t)num_envs, sizeof(int32_t));
    v->info_ret = (double *)calloc((size_t)num_envs, sizeof(double));
    v->info_score = (double *)calloc((size_t)num_envs, sizeof(double));
    for (int i = 0; i < num_envs; i++) v->envs[i].done = 1;
    return v;
}

void po_stochastic_free(po_stochastic_vec *v) {
    if (!v) return;
    free(v->envs); free(v->obs); free(v->rewards); free(v->terminals); free(v->truncations); free(v->masks);
    free(v->info_env); free(v->info_len); free(v->info_ret); free(v->info_score); free(v);
}

/* ocean.py:551-560 reset + emulation.py:169-192 + postprocess.py:18-20: observation [0.], nothing else to draw */
static void st_reset(po_stochastic_vec *v, int i) {
    st_env *e = &v->envs[i];
    e->tick = e->count = 0;
    e->done = 0;
    e->ep_return = 0.0;
    e->ep_length = 0;
    v->obs[i] = 0.0f;
    v->rewards[i] = 0.0f;
    v->terminals[i] = 0;
    v->truncations[i] = 0;
    v->masks[i] = 1;
}

/* ocean.py:562-582 */
double po_stochastic_reward(double p, int tick, int count, int action, double *proximity_out) {
    const double atn0_frac = (double)count / (double)tick;           /* python int / int */
    const double proximity = 1.0 - pow(p - atn0_frac, 2.0);          /* float ** 2 -> libm pow (CPython float_pow) */
    if (proximity_out) *proximity_out = proximity;
    return ((action == 0 && atn0_frac < p) || (action == 1 && atn0_frac >= p)) ? proximity : 0.0;
}

void po_stochastic_async_reset(po_stochastic_vec *v, int64_t seed) {
    (void)seed; /* reset(seed) only seeds the global RNGs, which this env never uses */
    v->n_infos = 0;
    for (int i = 0; i < v->n; i++) st_reset(v, i);
}

void po_stochastic_send(po_stochastic_vec *v, const int64_t *actions) {
    v->n_infos = 0;
    for (int i = 0; i < v->n; i++) {
        st_env *e = &v->envs[i];
        if (e->done) { /* vector.py:147-149: action ignored, reset row */
            st_reset(v, i);
            continue;
        }
        const int a = (int)actions[i];
        e->tick += 1;
        e->count += a == 0;
        double prox;
        const double reward = po_stochastic_reward(v->p, e->tick, e->count, a, &prox);
        const int terminal = e->tick == v->horizon;
        e->ep_return += reward;                 /* postprocess.py:31, summed in order at the end (:38) */
        e->ep_length += 1;
        v->obs[i] = 0.0f;
        v->rewards[i] = (float)reward;          /* emulation.py:219 buf.rewards[0] = reward */
        v->terminals[i] = (uint8_t)terminal;
        v->truncations[i] = 0;
        v->masks[i] = 1;
        e->done = terminal;                     /* emulation.py:226 */
        if (terminal) {
            const int k = v->n_infos++;
            v->info_env[k] = i;
            v->info_ret[k] = e->ep_return;
            v->info_len[k] = e->ep_length;
            v->info_score[k] = prox;
        }
    }
}

const float *po_stochastic_observations(const po_stochastic_vec *v) { return v->obs; }
const float *po_stochastic_rewards(const po_stochastic_vec *v) { return v->rewards; }
const uint8_t *po_stochastic_terminals(const po_stochastic_vec *v) { return v->terminals; }
const uint8_t *po_stochastic_truncations(const po_stochastic_vec *v) { return v->truncations; }
const uint8_t *po_stochastic_masks(const po_stochastic_vec *v) { return v->masks; }
int po_stochastic_num_infos(const po_stochastic_vec *v) { return v->n_infos; }
const int32_t *po_stochastic_info_env(const po_stochastic_vec *v) { return v->info_env; }
const double *po_stochastic_info_return(const po_stochastic_vec *v) { return v->info_ret; }
const int32_t *po_stochastic_info_length(const po_stochastic_vec *v) { return v->info_len; }
const double *po_stochastic_info_score(const po_stochastic_vec *v) { return v->info_score; }

/* ======================================================================================== */
/* GAE — c_gae.pyx:11-32                                                                      */
/* ======================================================================================== */
void po_compute_gae(const float *dones, const float *values, const float *rewards,
                    float *advantages, int num_steps, float gamma, float gae_lambda) {
    for (int i = 0; i < num_steps; i++) advantages[i] = 0.0f; /* np.zeros, :15 */
    float lastgaelam = 0.0f;
    for (int t = 0; t < num_steps - 1; t++) {
        int t_cur = num_steps - 2 - t;
        int t_next = num_steps - 1 - t;
        float nextnonterminal = 1.0f - dones[t_next];
        float delta = rewards[t_next] + gamma * values[t_next] * nextnonterminal - values[t_cur];
        lastgaelam = delta + gamma * gae_lambda * nextnonterminal * lastgaelam;
        advantages[t_cur] = lastgaelam;
    }
}

/* ======================================================================================== */
/* Philox4x32-10 (Salmon et al., SC'11) — our own noise definition, restated for checking     */
/* ======================================================================================== */
void po_philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
    uint32_t k0 = key_in[0], k1 = key_in[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
