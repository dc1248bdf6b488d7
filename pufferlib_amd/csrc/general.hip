// general.hip — the pieces of the WIDTH-GENERAL policy path (pufferlib_amd/general.py) that are not GEMMs.
//
// The fused kernels (rollout.hip, ppo_update.hip, lstm_*.hip) are instantiated for pufferlib.models.Default(hidden_size=128) /
// LSTMWrapper(128, 128) with up to 15 logits — the reference's defaults (models.py:24,65).  Any other width runs as a sequence
// of fp32-MFMA GEMM launches (csrc/igemm.hip: pfa_igemm_rows / pfa_igemm_weights — encoder, heads, the LSTM's gate product
// [x | h] Wcat^T and their transposes) plus the row-wise kernels here:
//   heads_rows_sample  sample_logits, action=None (frameworks/cleanrl.py:25-47) on precomputed head outputs [rows][NO]
//                      (columns < A the logits of all heads, column A the value): action, log-prob, entropy, value
//   heads_rows_eval    sample_logits with GIVEN actions (cleanrl.py:38-44): log-prob, entropy, value — the training-mode call
//                      policy(obs, action=...) of frameworks.cleanrl.Policy / RecurrentPolicy (cleanrl.py:60-66,87-93)
//   heads_rows_loss    the PPO loss of clean_pufferl.py:202-238 for the rows of a chunk and d loss / d (head outputs)
//   lstm_cell_fwd/bwd  the element-wise part of one nn.LSTM step (gate order i, f, g, o: models.py:76) and of its
//                      back-propagation through time
// One thread per row: up to 63 logits in up to 8 MultiDiscrete heads (one Discrete head may use all 63), reductions as plain
// loops in index order.  This path is about coverage, not the roofline: the headline workload never takes it.
#include <cmath>

#include "common.hpp"
#include "philox.hpp"
#include "ppo_tile.hpp"
#include "sampler.hpp"

namespace pfa {

constexpr int kGenMaxOut = 64;

struct HeadSpec {   // a = logits in total; nibble-packed head sizes (0: one Discrete(a) head, a <= 63)
    int a;
    uint32_t heads;
    __device__ __forceinline__ int count() const {
        if (heads == 0) return 1;
        int n = 0;
        while (n < 8 && ((heads >> (4 * n)) & 15u)) ++n;
        return n;
    }
    __device__ __forceinline__ int size(int h) const { return heads == 0 ? a : (int)((heads >> (4 * h)) & 15u); }
};

// rows of a chunk -> minibatch row: minibatch order (R == 0) or time-major chunks (row = t * R + k  <->  q0 + k * Th + t)
__device__ __forceinline__ long long chunk_row_to_q(long long row, long long q0, int R, int Th) {
    if (R <= 0) return q0 + row;
    const long long t = row / R, k = row - t * R;
    return q0 + k * Th + t;
}

__global__ void __launch_bounds__(256) heads_rows_sample_kernel(const float *out, int ld, long long rows, HeadSpec hs, const float *noise,
                                                               uint64_t seed, uint64_t step, long long row_offset, long long *actions,
                                                               float *logprob, float *entropy, float *value) {
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const float *o = out + row * ld;
    const float *nz = noise ? noise + row * hs.a : nullptr;
    long long packed = 0;
    float lp = 0.0f, ent = 0.0f;
    int start = 0;
    const int nh = hs.count();
    for (int h = 0; h < nh; ++h) {
        const int sz = hs.size(h);
        float mx = -INFINITY;
        for (int j = 0; j < sz; ++j) mx = fmaxf(mx, o[start + j]);
        float se = 0.0f;
        for (int j = 0; j < sz; ++j) se += expf(o[start + j] - mx);
        const float lse = mx + logf(se);
        float best = -INFINITY, he = 0.0f;
        int besti = 0;
        for (int j = 0; j < sz; ++j) {
            const float l = o[start + j];
            const float q = noise_lane(nz, seed, step, (uint64_t)(row_offset + row), start + j, hs.a);
            const float score = (expf(l - mx) / se) / q;   // torch.multinomial == argmax(p / q), first index wins ties
            if (score > best) {
                best = score;
                besti = j;
            }
            const float nl = l - lse;
            he += -nl * expf(nl);
        }
        lp += o[start + besti] - lse;
        ent += he;
        packed |= (long long)besti << (hs.heads == 0 ? 0 : 4 * h);
        start += sz;
    }
    actions[row] = packed;
    logprob[row] = lp;
    if (entropy) entropy[row] = ent;
    value[row] = o[hs.a];
}

__global__ void __launch_bounds__(256) heads_rows_eval_kernel(const float *out, int ld, long long rows, HeadSpec hs, const long long *actions,
                                                             float *logprob, float *entropy, float *value) {
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const float *o = out + row * ld;
    const long long packed = actions[row];
    float lp = 0.0f, ent = 0.0f;
    int start = 0;
    const int nh = hs.count();
    for (int h = 0; h < nh; ++h) {
        const int sz = hs.size(h);
        float mx = -INFINITY;
        for (int j = 0; j < sz; ++j) mx = fmaxf(mx, o[start + j]);
        float se = 0.0f;
        for (int j = 0; j < sz; ++j) se += expf(o[start + j] - mx);
        const float lse = mx + logf(se);
        float he = 0.0f;
        for (int j = 0; j < sz; ++j) {
            const float nl = o[start + j] - lse;
            he += -nl * (expf(o[start + j] - mx) / se);
        }
        const int act = hs.heads == 0 ? (int)packed : (int)((packed >> (4 * h)) & 15);
        lp += o[start + (act < sz && act >= 0 ? act : 0)] - lse;
        ent += he;
        start += sz;
    }
    logprob[row] = lp;
    entropy[row] = ent;
    value[row] = o[hs.a];
}

__global__ void __launch_bounds__(256) heads_rows_loss_kernel(const float *out, int ld, long long rows, RowMap map, long long q0, int R,
                                                             pfa_experience ex, HeadSpec hs, pfa_ppo_hparams hp, const double *adv_stats,
                                                             double global_rows, float *dout, int ldd, int no,
                                                             double *stats_partial /* [gridDim.x][8] */) {
    __shared__ double st[4][8];
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    if (row < rows) {
        float adv_mean = 0.0f, adv_den = 1.0f;
        if (hp.norm_adv) {   // clean_pufferl.py:211-213: unbiased std over the GLOBAL minibatch
            const double s1 = adv_stats[2 * map.mb], s2 = adv_stats[2 * map.mb + 1];
            const double mean = s1 / global_rows;
            double var = (s2 - s1 * mean) / (global_rows - 1.0);
            var = var > 0.0 ? var : 0.0;
            adv_mean = (float)mean;
            adv_den = (float)sqrt(var) + 1e-8f;
        }
        const float inv_rows = (float)(1.0 / global_rows);
        const long long fr = map.flat(chunk_row_to_q(row, q0, R, hp.bptt_horizon));
        const int packed = ex.actions[fr];
        const float old_logprob = ex.logprobs[fr], old_value = ex.values[fr], adv_raw = ex.advantages[fr], ret = ex.returns[fr];
        const float *o = out + row * ld;
        float *d = dout + row * ldd;
        // pass 1: per head log-sum-exp, entropy, chosen log-probability (cleanrl.py:38-44)
        float lse_h[8], ent_h[8], new_logprob = 0.0f, ent = 0.0f;
        int act_h[8];
        const int nh = hs.count();
        int start = 0;
        for (int h = 0; h < nh; ++h) {
            const int sz = hs.size(h);
            float mx = -INFINITY;
            for (int j = 0; j < sz; ++j) mx = fmaxf(mx, o[start + j]);
            float se = 0.0f;
            for (int j = 0; j < sz; ++j) se += expf(o[start + j] - mx);
            const float lse = mx + logf(se);
            float he = 0.0f;
            for (int j = 0; j < sz; ++j) {
                const float nl = o[start + j] - lse;
                he += -nl * (expf(o[start + j] - mx) / se);
            }
            int act = hs.heads == 0 ? packed : ((packed >> (4 * h)) & 15);
            act = act >= 0 && act < sz ? act : 0;
            new_logprob += o[start + act] - lse;
            lse_h[h] = lse;
            ent_h[h] = he;
            act_h[h] = start + act;
            ent += he;
            start += sz;
        }
        const float new_value = o[hs.a];
        const float logratio = new_logprob - old_logprob;
        const float ratio = expf(logratio);
        const float adv = hp.norm_adv ? (adv_raw - adv_mean) / adv_den : adv_raw;
        const float lo_c = 1.0f - hp.clip_coef, hi_c = 1.0f + hp.clip_coef;
        const float pg1 = -adv * ratio, pg2 = -adv * fminf(fmaxf(ratio, lo_c), hi_c);
        const bool inside = ratio >= lo_c && ratio <= hi_c;
        float dpg;   // torch.max's tie rule + clamp's pass-through, as in ppo_tile.hpp
        if (pg1 > pg2) dpg = -adv;
        else if (pg1 < pg2) dpg = inside ? -adv : 0.0f;
        else dpg = inside ? -adv : -0.5f * adv;
        const float scale = inv_rows;
        const float g_lp = dpg * ratio * scale;
        float v_loss, dv;
        if (hp.clip_vloss) {
            const float du = new_value - ret, vl_u = du * du;
            const float delta = new_value - old_value;
            const float vcl = old_value + fminf(fmaxf(delta, -hp.vf_clip_coef), hp.vf_clip_coef);
            const float dc = vcl - ret, vl_c = dc * dc;
            const bool vin = delta >= -hp.vf_clip_coef && delta <= hp.vf_clip_coef;
            v_loss = 0.5f * fmaxf(vl_u, vl_c);
            const float gu = 2.0f * du, gc = vin ? 2.0f * dc : 0.0f;
            dv = 0.5f * (vl_u > vl_c ? gu : (vl_u < vl_c ? gc : 0.5f * (gu + gc)));
        } else {
            const float du = new_value - ret;
            v_loss = 0.5f * du * du;
            dv = du;
        }
        dv *= hp.vf_coef * scale;
        // pass 2: d loss / d out
        start = 0;
        for (int h = 0; h < nh; ++h) {
            const int sz = hs.size(h);
            // (the head's exponentials again: cheaper than 63 registers per thread)
            float mx = -INFINITY;
            for (int j = 0; j < sz; ++j) mx = fmaxf(mx, o[start + j]);
            float se = 0.0f;
            for (int j = 0; j < sz; ++j) se += expf(o[start + j] - mx);
            for (int j = 0; j < sz; ++j) {
                const float nl = o[start + j] - lse_h[h], p = expf(o[start + j] - mx) / se;
                // d new_logprob/d logit = [chosen] - p ; d entropy/d logit = -p (nl + H_head)
                d[start + j] = g_lp * ((start + j == act_h[h] ? 1.0f : 0.0f) - p) + hp.ent_coef * scale * p * (nl + ent_h[h]);
            }
            start += sz;
        }
        d[hs.a] = dv;
        for (int j = hs.a + 1; j < no; ++j) d[j] = 0.0f;
        acc[0] = (double)fmaxf(pg1, pg2);
        acc[1] = (double)v_loss;
        acc[2] = (double)ent;
        acc[3] = (double)(-logratio);
        acc[4] = (double)((ratio - 1.0f) - logratio);
        acc[5] = (double)(fabsf(ratio - 1.0f) > hp.clip_coef ? 1.0f : 0.0f);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) acc[i] += __shfl_xor(acc[i], s, 64);   // fixed order
    if (lane_id() == 0)
        for (int i = 0; i < 8; ++i) st[wave_id()][i] = i < 6 ? acc[i] : 0.0;
    __syncthreads();
    if (threadIdx.x < 8) stats_partial[(size_t)blockIdx.x * 8 + threadIdx.x] = (st[0][threadIdx.x] + st[1][threadIdx.x]) + (st[2][threadIdx.x] + st[3][threadIdx.x]);
}

// loss_pairs16 (+)= the (hi, lo) float pairs of the chunk's six f64 sums, fixed summation order over the block partials
__global__ void __launch_bounds__(256) gen_stats_final_kernel(const double *partial, int nblocks, float *loss_pairs16, int accumulate) {
    __shared__ double sh[32][8];      // 32 strided chains per statistic, combined in a fixed order
    const int i = threadIdx.x & 7, part = threadIdx.x >> 3;
    double s = 0.0;
    for (int b = part; b < nblocks; b += 32) s += partial[(size_t)b * 8 + i];
    sh[part][i] = s;
    __syncthreads();
    if (threadIdx.x >= 8) return;
    s = 0.0;
    for (int q = 0; q < 32; ++q) s += sh[q][i];
    if (accumulate) s += (double)loss_pairs16[2 * i] + (double)loss_pairs16[2 * i + 1];
    const float hi = (float)s;
    loss_pairs16[2 * i] = hi;
    loss_pairs16[2 * i + 1] = (float)(s - (double)hi);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// One nn.LSTM step, element-wise part.  G [R][4H] holds W_ih x + b_ih + W_hh h + b_hh (gate order i, f, g, o) and is replaced by
// the ACTIVATED gates (what the backward needs).  c_out / h_out may alias nothing else; h_out2 (nullable) is a second copy of
// h (the [x | h] operand row of the next step).
__global__ void __launch_bounds__(256) lstm_cell_fwd_kernel(float *G, const float *c_prev, float *c_out, float *h_out, int ldh, float *h_out2,
                                                           int ldh2, long long R, int H) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * H) return;
    const long long r = i / H;
    const int u = (int)(i - r * H);
    float *g = G + r * 4 * H;
    const float ig = sigmoidf_(g[u]), fg = sigmoidf_(g[H + u]), gg = tanhf(g[2 * H + u]), og = sigmoidf_(g[3 * H + u]);
    const float c = fg * c_prev[i] + ig * gg;
    const float h = og * tanhf(c);
    g[u] = ig;
    g[H + u] = fg;
    g[2 * H + u] = gg;
    g[3 * H + u] = og;
    c_out[i] = c;
    h_out[r * ldh + u] = h;
    if (h_out2) h_out2[r * ldh2 + u] = h;
}

// Back-propagation through one step: dh = dh_a (+ dh_b), dc = the running cell gradient (in: from step t+1, out: for step t-1),
// Gact the activated gates of step t, c_prev / c the cell state before / after it.  dG [R][4H] = d loss / d (pre-activation gates).
__global__ void __launch_bounds__(256) lstm_cell_bwd_kernel(const float *dh_a, int lda, const float *dh_b, int ldb, float *dc, const float *Gact,
                                                           const float *c_prev, const float *c, float *dG, long long R, int H) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * H) return;
    const long long r = i / H;
    const int u = (int)(i - r * H);
    const float *g = Gact + r * 4 * H;
    const float ig = g[u], fg = g[H + u], gg = g[2 * H + u], og = g[3 * H + u];
    float dh = dh_a[r * lda + u];
    if (dh_b) dh += dh_b[r * ldb + u];
    const float tc = tanhf(c[i]);
    const float dct = dc[i] + dh * og * (1.0f - tc * tc);
    float *d = dG + r * 4 * H;
    d[u] = dct * gg * ig * (1.0f - ig);
    d[H + u] = dct * c_prev[i] * fg * (1.0f - fg);
    d[2 * H + u] = dct * ig * (1.0f - gg * gg);
    d[3 * H + u] = dh * tc * og * (1.0f - og);
    dc[i] = dct * fg;
}

// Row-block copies between the two row orders of a chunk of Rc segments x Th steps — segment-major (row k * Th + t: how a
// minibatch is stored and how the conv encoder batches frames) and time-major (row t * Rc + k: every step a contiguous block,
// what the LSTM walks) — with an optional relu' mask from the post-ReLU activation the gradient belongs to.
//   to_time_major != 0: dst[t * R + k] = src[k * Th + t]        else: dst[k * Th + t] = src[t * R + k]
//   act (nullable, laid out like SRC with row stride lda): dst = act > 0 ? src : 0.   R == 0: plain row-for-row copy / mask.
__global__ void __launch_bounds__(256) rows_perm_kernel(const float *src, int lds_, float *dst, int ldd, const float *act, int lda,
                                                       long long rows, int cols, int R, int Th, int to_time_major) {
    const int c4 = cols / 4;
    const long long total = rows * c4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / c4;
        const int c = (int)(i - r * c4) * 4;
        long long rd = r;
        if (R > 0) {
            if (to_time_major) {   // r = k * Th + t
                const long long k = r / Th, t = r - k * Th;
                rd = t * R + k;
            } else {               // r = t * R + k
                const long long t = r / R, k = r - t * R;
                rd = k * Th + t;
            }
        }
        float4 v = *reinterpret_cast<const float4 *>(src + r * lds_ + c);
        if (act) {
            const float4 a4 = *reinterpret_cast<const float4 *>(act + r * lda + c);
            v.x = a4.x > 0.0f ? v.x : 0.0f;
            v.y = a4.y > 0.0f ? v.y : 0.0f;
            v.z = a4.z > 0.0f ? v.z : 0.0f;
            v.w = a4.w > 0.0f ? v.w : 0.0f;
        }
        *reinterpret_cast<float4 *>(dst + rd * ldd + c) = v;
    }
}

static int check_heads(int32_t a, uint32_t heads, int32_t ld) {
    PFA_REQUIRE(a >= 1 && a < kGenMaxOut && ld > a, "heads: 1..63 logits, and the value in column num_actions of a row of ld > num_actions floats");
    if (heads != 0) {
        int n = 0, total = 0;
        for (; n < 8 && ((heads >> (4 * n)) & 15u) != 0; ++n) total += (int)((heads >> (4 * n)) & 15u);
        PFA_REQUIRE(total == a && (n == 8 || (heads >> (4 * n)) == 0), "heads: head sizes 0x%x do not sum to num_actions %d", heads, a);
    }
    return 0;
}

}  // namespace pfa

using namespace pfa;

extern "C" int pfa_heads_rows_sample(const float *out, int32_t ld, int64_t rows, int32_t num_actions, uint32_t heads, const float *noise,
                                     const pfa_noise_key *key, int64_t row_offset, int64_t *actions, float *logprob, float *entropy,
                                     float *value, pfa_stream_t stream) {
    PFA_REQUIRE(out && actions && logprob && value && rows >= 0, "heads_rows_sample: null buffer");
    PFA_REQUIRE(noise || key, "heads_rows_sample: need an explicit noise tensor or a Philox key");
    if (int rc = check_heads(num_actions, heads, ld)) return rc;
    if (rows == 0) return 0;
    ScopedKernelTimer timer("heads_rows_sample", (hipStream_t)stream);
    hipLaunchKernelGGL(heads_rows_sample_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, (int)ld,
                       (long long)rows, HeadSpec{(int)num_actions, heads}, noise, key ? key->seed : 0, key ? key->step : 0,
                       (long long)row_offset, (long long *)actions, logprob, entropy, value);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_heads_rows_eval(const float *out, int32_t ld, int64_t rows, int32_t num_actions, uint32_t heads, const int64_t *actions,
                                   float *logprob, float *entropy, float *value, pfa_stream_t stream) {
    PFA_REQUIRE(out && actions && logprob && entropy && value && rows >= 0, "heads_rows_eval: null buffer");
    if (int rc = check_heads(num_actions, heads, ld)) return rc;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(heads_rows_eval_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, (int)ld,
                       (long long)rows, HeadSpec{(int)num_actions, heads}, (const long long *)actions, logprob, entropy, value);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t pfa_heads_rows_loss_workspace_bytes(int64_t rows) { return (size_t)((rows + 255) / 256 + 1) * 8 * sizeof(double); }

extern "C" int pfa_heads_rows_loss(const float *out, int32_t ld, const pfa_experience *exp, int64_t batch_rows, int32_t mb, int64_t q0,
                                   int64_t rows, int32_t time_major_rows, int32_t num_actions, uint32_t heads, const pfa_ppo_hparams *hp,
                                   const double *adv_stats, int64_t global_mb_rows, float *dout, int32_t ldd, int32_t num_out,
                                   float *loss_pairs16, int32_t accumulate, void *workspace, pfa_stream_t stream) {
    PFA_REQUIRE(out && exp && hp && dout && loss_pairs16 && workspace, "heads_rows_loss: null buffer");
    PFA_REQUIRE(exp->actions && exp->logprobs && exp->values && exp->advantages && exp->returns, "heads_rows_loss: null experience buffer");
    if (int rc = check_heads(num_actions, heads, ld)) return rc;
    PFA_REQUIRE(num_out > num_actions && num_out <= kGenMaxOut && ldd >= num_out, "heads_rows_loss: num_out must cover the value column");
    PFA_REQUIRE(hp->num_minibatches >= 1 && hp->bptt_horizon >= 1 && batch_rows % hp->num_minibatches == 0, "heads_rows_loss: bad minibatch partition");
    const int64_t mbs = batch_rows / hp->num_minibatches;
    PFA_REQUIRE(mb >= 0 && mb < hp->num_minibatches && q0 >= 0 && rows >= 1 && q0 + rows <= mbs, "heads_rows_loss: chunk outside the minibatch");
    PFA_REQUIRE(time_major_rows == 0 || (rows % time_major_rows == 0 && rows / time_major_rows == hp->bptt_horizon && q0 % hp->bptt_horizon == 0),
                "heads_rows_loss: a time-major chunk is [bptt_horizon][time_major_rows] whole segments");
    PFA_REQUIRE(!hp->norm_adv || adv_stats, "heads_rows_loss: norm_adv needs adv_stats");
    RowMap map{mb, hp->num_minibatches, hp->bptt_horizon};
    const unsigned grid = (unsigned)((rows + 255) / 256);
    hipLaunchKernelGGL(heads_rows_loss_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, (int)ld, (long long)rows, map, (long long)q0,
                       (int)time_major_rows, *exp, HeadSpec{(int)num_actions, heads}, *hp, adv_stats, (double)global_mb_rows, dout, (int)ldd,
                       (int)num_out, (double *)workspace);
    PFA_LAUNCH_CHECK();
    hipLaunchKernelGGL(gen_stats_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double *)workspace, (int)grid, loss_pairs16,
                       (int)accumulate);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_lstm_cell_forward(float *gates, const float *c_prev, float *c_out, float *h_out, int32_t ldh, float *h_out2, int32_t ldh2,
                                     int64_t rows, int32_t hidden, pfa_stream_t stream) {
    PFA_REQUIRE(gates && c_prev && c_out && h_out && rows >= 0 && hidden >= 1 && ldh >= hidden && (!h_out2 || ldh2 >= hidden), "lstm_cell_forward: bad arguments");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3((unsigned)((rows * hidden + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gates, c_prev, c_out,
                       h_out, (int)ldh, h_out2, (int)ldh2, (long long)rows, (int)hidden);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_lstm_cell_backward(const float *dh_a, int32_t lda, const float *dh_b, int32_t ldb, float *dc, const float *gates_act,
                                      const float *c_prev, const float *c, float *dgates, int64_t rows, int32_t hidden, pfa_stream_t stream) {
    PFA_REQUIRE(dh_a && dc && gates_act && c_prev && c && dgates && rows >= 0 && hidden >= 1 && lda >= hidden && (!dh_b || ldb >= hidden),
                "lstm_cell_backward: bad arguments");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3((unsigned)((rows * hidden + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dh_a, (int)lda, dh_b,
                       (int)ldb, dc, gates_act, c_prev, c, dgates, (long long)rows, (int)hidden);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_rows_perm(const float *src, int32_t lds_, float *dst, int32_t ldd, const float *act, int32_t lda, int64_t rows, int32_t cols,
                             int32_t segments, int32_t steps, int32_t to_time_major, pfa_stream_t stream) {
    PFA_REQUIRE(src && dst && rows >= 0 && cols >= 4 && cols % 4 == 0 && lds_ >= cols && ldd >= cols && lds_ % 4 == 0 && ldd % 4 == 0 &&
                    (!act || (lda >= cols && lda % 4 == 0)),
                "rows_perm: rows of whole float4s with 16-byte aligned strides");
    PFA_REQUIRE((((uintptr_t)src | (uintptr_t)dst | (uintptr_t)act) & 15) == 0, "rows_perm: buffers must be 16-byte aligned");
    PFA_REQUIRE(segments == 0 || (steps >= 1 && (int64_t)segments * steps == rows), "rows_perm: rows != segments x steps");
    if (rows == 0) return 0;
    const long long total = rows * (cols / 4);
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(rows_perm_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, src, (int)lds_, dst,
                       (int)ldd, act, (int)lda, (long long)rows, (int)cols, (int)segments, (int)steps, (int)to_time_major);
    PFA_LAUNCH_CHECK();
    return 0;
}
