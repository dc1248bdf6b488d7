// spaces.hip — ocean `Spaces` (pufferlib/environments/ocean/ocean.py:356-404) as a device-resident vecenv: the env with a
// structured (Dict) observation and a structured (Dict) action.  Reference stack per env: pufferlib.vector.Serial
// (vector.py:78-162) over make_spaces (ocean/environment.py:66-69) = GymnasiumPufferEnv + EpisodeStats + ocean.Spaces, i.e.
// observations emulated into 108-byte rows {flat: int8[5] @0, image: f32[5][5] @8} (emulation.py:68-110, align=True) which
// models.Default reads byte by byte as floats (models.py:50), actions MultiDiscrete([2, 2]) = (flat, image) (Dict keys sorted,
// emulation.py:111-121).  Every step is terminal: reward = 0.5 [image action == (sum(image) > 0)] + 0.5 [flat action ==
// (sum(flat) > 0)], info score = reward; the next send() is the auto-reset row.
//
// Randomness: Spaces.reset ignores its seed and draws from numpy's PROCESS-GLOBAL legacy generator (ocean.py:380-383):
// randn(5, 5) — 25 legacy_gauss values: a polar attempt takes two 53-bit doubles = 4 MT19937 words and is rejected when
// r2 >= 1 or r2 == 0; an accepted attempt yields TWO values, the second stays cached for the next call — then
// randint(-1, 2, (5,), dtype=int8): bytes of buffered 32-bit words, low byte first, a byte b is accepted when (b & 3) <= 2, the
// word buffer is dropped at the end of the call.  The generator is seeded by clean_pufferl.seed_everything (np.random.seed,
// clean_pufferl.py:596-600); async_reset(seed) here restates that call followed by the N initial resets.  A reset consumes a
// DATA-DEPENDENT number of words, and whether it starts with a cached gaussian alternates (25 is odd), so the position of
// reset k in the stream depends on every reset before it.  The stream does not depend on actions: like the Squared and Memory
// tapes it is resolved ahead of time, here in its parallel form (oracle/spaces_stream.py is the numpy prototype, pinned against
// the sequential C restatement and through it against the reference):
//   1. acc[p]     would a polar attempt starting at word p be accepted?                                     (every p in parallel)
//   2. len[p][c]  words one reset consumes when it starts at p with (c = 1) / without a cached gaussian: walk the marks
//                 for 13 - c accepted attempts, then the byte loop                                          (every (p, c) in parallel)
//   3. the chain start(k+1) = start(k) + len[start(k)][c(k)], c(k+1) = 1 - c(k): one thread, a few hundred LDS reads
//   4. every reset of the window computes its row from its own start                                        (a thread per reset)
// over windows of 16 MT19937 blocks held in LDS by ONE workgroup (the 624-word recurrence is sequential between blocks).
#include <cmath>

#include "common.hpp"
#include "episode_fin.hpp"
#include "mt19937.hpp"

namespace pfa {

constexpr int kSpDP = 128;          // observation row stride in floats (108 byte values + padding)
constexpr int kSpRow = 108;         // emulated row bytes
constexpr int kSpTapeRow = 112;     // tape row: the 108 bytes, then image_sign, flat_sign, 2 pad
constexpr int kSpThreads = 1024;
constexpr int kSpWindow = 16 * kMtN;   // words held in LDS
constexpr int kSpMaxResets = 512;      // resets resolved per window (a reset takes >= 50 words)

struct SpacesHeader {
    uint32_t mt[kMtN];       // raw MT19937 state behind the last generated block
    uint32_t carry[kSpWindow];   // generated (tempered) words not yet consumed by a reset
    int carry_count;
    int has_gauss;           // legacy_gauss's cache, carried between fill_tape calls
    double gauss;
    int underrun, error;     // error: a window could not resolve a single reset (cannot happen with 16 blocks)
    long long rounds_filled;
};
struct SpacesEnv {
    int done, image_sign, flat_sign, pad;
    long long rounds;        // tape rounds this env has consumed (round 0 = the async_reset observation)
};
struct SpacesView {
    SpacesHeader *hdr;
    SpacesEnv *env;
    EpisodeFin *fin;
    uint8_t *tape;           // [tape_rounds][n][kSpTapeRow]
    int n, tape_rounds;
};
__host__ __device__ inline size_t spaces_state_bytes(int n, int tape_rounds) {
    return (sizeof(SpacesHeader) + 15) / 16 * 16 + (size_t)n * (sizeof(SpacesEnv) + sizeof(EpisodeFin)) + (size_t)tape_rounds * n * kSpTapeRow;
}
__host__ __device__ inline SpacesView spaces_view(void *state, const pfa_spaces_config &c) {
    SpacesView v;
    char *p = (char *)state;
    v.hdr = (SpacesHeader *)p;
    p += (sizeof(SpacesHeader) + 15) / 16 * 16;
    v.env = (SpacesEnv *)p;
    p += (size_t)c.num_envs * sizeof(SpacesEnv);
    v.fin = (EpisodeFin *)p;
    p += (size_t)c.num_envs * sizeof(EpisodeFin);
    v.tape = (uint8_t *)p;
    v.n = c.num_envs;
    v.tape_rounds = c.tape_rounds;
    return v;
}

// np.random.seed(seed): mt19937_seed = init_genrand, the cached gaussian cleared (numpy/random/_legacy: _legacy_seeding)
__global__ void spaces_seed_kernel(SpacesView v, uint32_t seed) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t x = seed;
    v.hdr->mt[0] = x;
    for (int i = 1; i < kMtN; ++i) {
        x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
        v.hdr->mt[i] = x;
    }
    v.hdr->carry_count = 0;
    v.hdr->has_gauss = 0;
    v.hdr->gauss = 0.0;
    v.hdr->underrun = 0;
    v.hdr->error = 0;
    v.hdr->rounds_filled = 0;
}

// mt19937_next_double on words p, p+1 (numpy/random/src/mt19937/mt19937.h)
__device__ __forceinline__ double sp_double(const uint32_t *w, int p) {
    return ((double)(w[p] >> 5) * 67108864.0 + (double)(w[p + 1] >> 6)) / 9007199254740992.0;
}
// one polar attempt of legacy_gauss on words p .. p+3: accepted? (and, for the caller that needs them, the two values)
__device__ __forceinline__ bool sp_attempt(const uint32_t *w, int p, double &first, double &second) {
#pragma clang fp contract(off)
    const double x1 = 2.0 * sp_double(w, p) - 1.0;
    const double x2 = 2.0 * sp_double(w, p + 2) - 1.0;
    const double r2 = x1 * x1 + x2 * x2;
    if (r2 >= 1.0 || r2 == 0.0) return false;
    const double f = sqrt(-2.0 * log(r2) / r2);
    first = f * x2;    // returned by this call
    second = f * x1;   // cached for the next one
    return true;
}
__device__ __forceinline__ bool sp_accepted(const uint32_t *w, int p) {
#pragma clang fp contract(off)
    const double x1 = 2.0 * sp_double(w, p) - 1.0;
    const double x2 = 2.0 * sp_double(w, p + 2) - 1.0;
    const double r2 = x1 * x1 + x2 * x2;
    return !(r2 >= 1.0 || r2 == 0.0);
}

// Draw `rounds` more reset rounds (rounds * N resets in env order) into the tape ring.  One workgroup.
__global__ void __launch_bounds__(kSpThreads) spaces_tape_kernel(SpacesView v, int rounds) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sp_lds[];
    uint32_t *w = sp_lds;                                   // [kSpWindow + 8] tempered words of the window
    uint32_t *mt = w + kSpWindow + 8;                       // [2][624] raw state, double-buffered
    uint32_t *blk = mt + 2 * kMtN;                          // [624] tempered output of one regeneration
    uint16_t *len = reinterpret_cast<uint16_t *>(blk + kMtN);   // [kSpWindow][2] words per reset, 0 = does not fit the window
    uint8_t *acc = reinterpret_cast<uint8_t *>(len + 2 * kSpWindow);   // [kSpWindow]
    int *rec_pos = reinterpret_cast<int *>(acc + kSpWindow);           // [kSpMaxResets] start word of reset i of this window
    double *rec_cache = reinterpret_cast<double *>(rec_pos + kSpMaxResets + 8);   // [kSpMaxResets] the gaussian reset i leaves cached
    __shared__ int s_nrec, s_newpos, s_c0;
    const int tid = threadIdx.x;
    const long long first_round = v.hdr->rounds_filled;
    const long long need = (long long)rounds * v.n;
    int avail = v.hdr->carry_count;
    for (int i = tid; i < avail; i += kSpThreads) w[i] = v.hdr->carry[i];
    if (tid < kMtN) mt[tid] = v.hdr->mt[tid];
    int cur = 0;
    int c0 = v.hdr->has_gauss;          // parity of the first unresolved reset
    double g0 = v.hdr->gauss;           // its cached gaussian (when c0)
    long long produced = 0;
    __syncthreads();
    while (produced < need) {
        while (avail + kMtN <= kSpWindow) {   // top the window up
            mt_next_block(mt + cur * kMtN, mt + (cur ^ 1) * kMtN, blk);
            cur ^= 1;
            if (tid < kMtN) w[avail + tid] = blk[tid];
            avail += kMtN;
            __syncthreads();
        }
        // 1. acceptance marks
        for (int p = tid; p < avail; p += kSpThreads) acc[p] = p + 3 < avail ? (sp_accepted(w, p) ? 1 : 0) : 0;
        __syncthreads();
        // 2. words per reset from every start and entry parity
        for (int i = tid; i < 2 * avail; i += kSpThreads) {
            const int p = i >> 1, c = i & 1;
            int q = p, k = 0;
            while (k < 13 - c && q + 3 < avail) {
                k += acc[q];
                q += 4;
            }
            int words = 0;
            if (k == 13 - c) {   // randint(-1, 2, 5, int8): accept a byte when (b & 3) <= 2; whole words are consumed
                int got = 0, examined = 0;
                while (got < 5 && q + (examined >> 2) < avail) {
                    const uint32_t b = (w[q + (examined >> 2)] >> (8 * (examined & 3))) & 0xFFu;
                    ++examined;
                    got += (b & 3u) <= 2u;
                }
                if (got == 5) words = (q - p) + ((examined + 3) >> 2);
            }
            len[i] = (uint16_t)words;
        }
        __syncthreads();
        // 3. the chain of reset starts
        if (tid == 0) {
            int pos = 0, c = c0, n = 0;
            const long long left = need - produced;
            while (n < kSpMaxResets && n < left) {
                const int l = len[2 * pos + c];
                if (l == 0) break;
                rec_pos[n++] = pos | (c << 30);
                pos += l;
                c ^= 1;
                if (pos >= avail) break;
            }
            s_nrec = n;
            s_newpos = pos;
            s_c0 = c;
        }
        __syncthreads();
        const int nrec = s_nrec;
        if (nrec == 0) {   // cannot happen with a 16-block window; never spin
            if (tid == 0) v.hdr->error = 1;
            break;
        }
        // 4. a thread per reset: the 25 gaussians (its cached one is patched in below), the 5 int8, the two signs
        float image[25];
        int k_reset = -1;
        if (tid < nrec) {
#pragma clang fp contract(off)
            k_reset = tid;
            const int p = rec_pos[tid] & 0x3FFFFFFF, c = rec_pos[tid] >> 30;
            int q = p, nv = c;   // slot 0 waits for the predecessor's cached value when c
            double cached = 0.0;
            while (nv < 25) {
                double first, second;
                if (sp_attempt(w, q, first, second)) {
                    image[nv++] = (float)first;
                    if (nv < 25) image[nv++] = (float)second;
                    else cached = second;
                }
                q += 4;
            }
            rec_cache[tid] = cached;   // meaningful when the NEXT reset starts with c = 1
            rec_pos[tid] = q | (c << 30);   // from here on: where the byte part starts
        }
        __syncthreads();
        if (k_reset >= 0) {
            const int c = rec_pos[k_reset] >> 30, q = rec_pos[k_reset] & 0x3FFFFFFF;
            if (c) image[0] = (float)(k_reset == 0 ? g0 : rec_cache[k_reset - 1]);
            const long long abs_reset = first_round * v.n + produced + k_reset;   // reset index since async_reset
            const long long round = abs_reset / v.n;
            const int env = (int)(abs_reset % v.n);
            uint8_t *row = v.tape + ((size_t)(round % v.tape_rounds) * v.n + env) * kSpTapeRow;
            int got = 0, examined = 0, fsum = 0;
            row[5] = row[6] = row[7] = 0;
            while (got < 5) {
                const uint32_t b = (w[q + (examined >> 2)] >> (8 * (examined & 3))) & 0xFFu;
                ++examined;
                if ((b & 3u) <= 2u) {
                    const uint8_t val = (uint8_t)(255u + (b & 3u));   // off = (uint8)(-1), wraps like numpy's C cast
                    row[got++] = val;
                    fsum += (int)(int8_t)val;
                }
            }
            for (int i = 0; i < 25; ++i) *reinterpret_cast<float *>(row + 8 + 4 * i) = image[i];
            // np.sum(float32[25]): pairwise sum with 8 accumulators over the first 24, then the tail (loops_utils.h.src)
            float r[8];
            for (int j = 0; j < 8; ++j) r[j] = image[j];
            for (int i = 8; i < 24; i += 8)
                for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], image[i + j]);
            float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])), __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
            res = __fadd_rn(res, image[24]);
            row[108] = res > 0.0f ? 1 : 0;
            row[109] = fsum > 0 ? 1 : 0;
            row[110] = row[111] = 0;
        }
        // carry the cache of the last resolved reset over to the next window
        const double g_next = rec_cache[nrec - 1];
        __syncthreads();
        g0 = g_next;
        c0 = s_c0;
        produced += nrec;
        // compact: unconsumed words to the front
        const int newpos = s_newpos < avail ? s_newpos : avail;
        const int rest = avail - newpos;
        uint32_t keep[(kSpWindow + kSpThreads - 1) / kSpThreads];
        int nk = 0;
        for (int i = tid; i < rest; i += kSpThreads) keep[nk++] = w[newpos + i];
        __syncthreads();
        nk = 0;
        for (int i = tid; i < rest; i += kSpThreads) w[i] = keep[nk++];
        avail = rest;
        __syncthreads();
    }
    __syncthreads();
    for (int i = tid; i < avail; i += kSpThreads) v.hdr->carry[i] = w[i];
    if (tid < kMtN) v.hdr->mt[tid] = mt[cur * kMtN + tid];
    if (tid == 0) {
        v.hdr->carry_count = avail;
        v.hdr->has_gauss = c0;
        v.hdr->gauss = g0;
        v.hdr->rounds_filled = first_round + rounds;
    }
}

__device__ __forceinline__ void spaces_load_row(const SpacesView &v, SpacesEnv &s, int e, float *obs) {
    if (s.rounds >= v.hdr->rounds_filled) v.hdr->underrun = 1;
    const uint8_t *row = v.tape + ((size_t)(s.rounds % v.tape_rounds) * v.n + e) * kSpTapeRow;
    float *o = obs + (size_t)e * kSpDP;
    for (int j = 0; j < kSpRow; ++j) o[j] = (float)row[j];   // models.Default: observations.float() of the raw bytes
    s.image_sign = row[108];
    s.flat_sign = row[109];
    s.done = 0;
    s.rounds += 1;
}

__global__ void __launch_bounds__(256) spaces_begin_kernel(SpacesView v, float *obs, float *rewards, uint8_t *terminals,
                                                          uint8_t *truncations, uint8_t *masks) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= v.n) return;
    SpacesEnv s = {};
    for (int j = kSpRow; j < kSpDP; ++j) obs[(size_t)e * kSpDP + j] = 0.0f;
    spaces_load_row(v, s, e, obs);
    v.env[e] = s;
    EpisodeFin f = {};
    v.fin[e] = f;
    rewards[e] = 0.0f;
    terminals[e] = 0;
    truncations[e] = 0;
    masks[e] = 1;
}

// actions: one packed word per env (pfa_mlp_dims.heads convention: head h in bits 4h..4h+3): head 0 = flat, head 1 = image
__global__ void __launch_bounds__(256) spaces_send_kernel(SpacesView v, const long long *actions, float *obs, float *rewards,
                                                         uint8_t *terminals, uint8_t *truncations, uint8_t *masks) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= v.n) return;
    SpacesEnv s = v.env[e];
    v.fin[e].last_fin = 0;
    float r = 0.0f;
    bool t = false;
    if (s.done) {   // the auto-reset row (vector.py:144-151): reward 0, terminal False, a fresh observation
        spaces_load_row(v, s, e, obs);
    } else {
        const int a_flat = (int)(actions[e] & 15), a_image = (int)((actions[e] >> 4) & 15);
        double reward = 0.0;
        if (s.image_sign == a_image) reward += 0.5;
        if (s.flat_sign == a_flat) reward += 0.5;
        r = (float)reward;
        t = true;
        s.done = 1;
        episode_account(v.fin[e], reward, 1, reward);   // EpisodeStats: return = the one reward, length 1, score = reward
    }
    v.env[e] = s;
    rewards[e] = r;
    terminals[e] = t ? 1 : 0;
    truncations[e] = 0;
    masks[e] = 1;
}

static int check_spaces_config(const pfa_spaces_config *c) {
    PFA_REQUIRE(c != nullptr, "spaces: null config");
    PFA_REQUIRE(c->num_envs >= 1, "spaces: num_envs must be >= 1");
    PFA_REQUIRE(c->tape_rounds >= 2, "spaces: tape_rounds must be >= 2");
    return 0;
}
static size_t spaces_tape_lds() {
    return (size_t)(kSpWindow + 8 + 3 * kMtN) * 4 + (size_t)2 * kSpWindow * 2 + kSpWindow + (kSpMaxResets + 8) * 4 + kSpMaxResets * 8 + 64;
}

}  // namespace pfa

using namespace pfa;

extern "C" size_t pfa_spaces_state_bytes(const pfa_spaces_config *cfg) {
    if (check_spaces_config(cfg)) return 0;
    return spaces_state_bytes(cfg->num_envs, cfg->tape_rounds);
}

extern "C" int pfa_spaces_fill_tape(void *state, const pfa_spaces_config *cfg, int32_t rounds, pfa_stream_t stream) {
    if (int rc = check_spaces_config(cfg)) return rc;
    PFA_REQUIRE(state && rounds >= 0 && rounds <= cfg->tape_rounds, "spaces.fill_tape: rounds must be in 0..tape_rounds");
    if (rounds == 0) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        PFA_CHECK_HIP(hipFuncSetAttribute((const void *)spaces_tape_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)spaces_tape_lds()));
        attr_set = true;
    }
    ScopedKernelTimer timer("spaces_tape", (hipStream_t)stream);
    hipLaunchKernelGGL(spaces_tape_kernel, dim3(1), dim3(kSpThreads), spaces_tape_lds(), (hipStream_t)stream, spaces_view(state, *cfg), (int)rounds);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_spaces_async_reset(void *state, const pfa_spaces_config *cfg, int64_t seed, float *obs, float *rewards,
                                      uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    if (int rc = check_spaces_config(cfg)) return rc;
    PFA_REQUIRE(state && obs && rewards && terminals && truncations && masks, "spaces.async_reset: null buffer");
    PFA_REQUIRE(seed >= 0 && seed <= 0xFFFFFFFFll, "spaces.async_reset: np.random.seed needs 0 <= seed < 2**32");
    hipLaunchKernelGGL(spaces_seed_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, spaces_view(state, *cfg), (uint32_t)seed);
    PFA_LAUNCH_CHECK();
    if (int rc = pfa_spaces_fill_tape(state, cfg, 1, stream)) return rc;   // round 0: the observations async_reset itself draws
    hipLaunchKernelGGL(spaces_begin_kernel, dim3((unsigned)((cfg->num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       spaces_view(state, *cfg), obs, rewards, terminals, truncations, masks);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_spaces_send(void *state, const pfa_spaces_config *cfg, const int64_t *actions, float *obs, float *rewards,
                               uint8_t *terminals, uint8_t *truncations, uint8_t *masks, pfa_stream_t stream) {
    if (int rc = check_spaces_config(cfg)) return rc;
    PFA_REQUIRE(state && actions && obs && rewards && terminals && truncations && masks, "spaces.send: null buffer");
    hipLaunchKernelGGL(spaces_send_kernel, dim3((unsigned)((cfg->num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       spaces_view(state, *cfg), (const long long *)actions, obs, rewards, terminals, truncations, masks);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_spaces_episode_stats(void *state, const pfa_spaces_config *cfg, double *out5, int32_t reset, pfa_stream_t stream) {
    if (int rc = check_spaces_config(cfg)) return rc;
    PFA_REQUIRE(state && out5, "spaces.episode_stats: null buffer");
    SpacesView v = spaces_view(state, *cfg);
    hipLaunchKernelGGL(episode_stats_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, v.fin, (int)cfg->num_envs, out5, (int)reset,
                       (const int *)&v.hdr->underrun);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_spaces_last_infos(void *state, const pfa_spaces_config *cfg, uint8_t *finished, double *episode_return,
                                     int32_t *episode_length, double *score, pfa_stream_t stream) {
    if (int rc = check_spaces_config(cfg)) return rc;
    PFA_REQUIRE(state && finished && episode_return && episode_length && score, "spaces.last_infos: null buffer");
    hipLaunchKernelGGL(episode_infos_kernel, dim3((unsigned)((cfg->num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       spaces_view(state, *cfg).fin, (int)cfg->num_envs, finished, episode_return, (int *)episode_length, score);
    PFA_LAUNCH_CHECK();
    return 0;
}
