// lstm.hip — kernels of the recurrent policy path: pufferlib.models.LSTMWrapper (pufferlib/models.py:64-111) =
// Default.encode_observations -> nn.LSTM(128, 128, 1) -> Default.decode_actions, wrapped by
// frameworks.cleanrl.RecurrentPolicy (frameworks/cleanrl.py:69-93), and its use in clean_pufferl.evaluate
// (clean_pufferl.py:100-105) and the BPTT minibatch loop of clean_pufferl.train (clean_pufferl.py:186-193).
//
// The products of this path are the fused MFMA kernels of lstm_fused.hip (rollout), lstm_seq.hip (training forward / BPTT) and
// gemm.hip (weight gradients).  This file holds the rest:
//   lstm_heads_loss                     decoder + value head, PPO loss, d loss/d heads, d loss/d h, loss sums, head bias gradient
//   gather_obs_time_major / store_rows / store_step / finish_grads / sumsq pieces for the clip norm
// and the step-wise pieces the fused kernels replaced, kept as protocol-level entry points and test anchors:
//   lstm_cell_fwd / lstm_cell_bwd      gate nonlinearities + cell/hidden update and their exact derivatives (gate order i,f,g,o)
//   lstm_heads_sample                   decoder + value head on h, sample_logits with the shared row16 sampler
//   relu / relu' / column sums
#include "common.hpp"
#include "lane_ops.hpp"
#include "lstm_tile.hpp"
#include "mlp_tile.hpp"
#include "sampler.hpp"

namespace pfa {

constexpr int kLstmH = 128;
constexpr int kHeadStatCols = 24;  // per-workgroup partials of lstm_heads_loss: 8 loss sums + 16 column sums of dout

// Heads weights of models.Default inside the flat parameter vector, staged in LDS as W2v[16][128(+1)].
__device__ __forceinline__ void stage_heads(const float *params, int dp, int a, float *w2v /* [16][129] */, float *b2v) {
    const MlpOffsets off = mlp_offsets(dp, a);
    for (int i = threadIdx.x; i < kOut * kLstmH; i += blockDim.x) {
        const int o = i / kLstmH, u = i - o * kLstmH;
        w2v[o * (kLstmH + 1) + u] = w2v_at(params, off, a, o, u);
    }
    for (int i = threadIdx.x; i < kOut; i += blockDim.x) b2v[i] = b2v_at(params, off, a, i);
}

// Experience.store of one rollout step (clean_pufferl.py:436-450) into the env-major buffers: row (e, t) at e*T + t.
__global__ void __launch_bounds__(256) store_step_kernel(pfa_experience ex, int t, int num_envs, int dp, const float *obs,
                                                        const float *rewards, const uint8_t *terminals, const long long *actions,
                                                        const float *logprob, const float *value) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int V = dp / 4;
    if (i < (long long)num_envs * V) {
        const int e = (int)(i / V), c4 = (int)(i - (long long)e * V);
        *reinterpret_cast<float4 *>(ex.obs + ((size_t)e * ex.horizon_T + t) * dp + 4 * c4) =
            *reinterpret_cast<const float4 *>(obs + (size_t)e * dp + 4 * c4);
    }
    if (i < num_envs) {
        const size_t row = (size_t)i * ex.horizon_T + t;
        ex.rewards[row] = rewards[i];
        ex.dones[row] = terminals[i] ? 1.0f : 0.0f;
        ex.actions[row] = (int)actions[i];
        ex.logprobs[row] = logprob[i];
        ex.values[row] = value[i];
    }
}

// Experience.store for ARBITRARY arrival order (clean_pufferl.py:436-450) fused with sort_training_data (:452-464): the
// reference appends rows as they arrive and later sorts them by (env_id, step); here row i of the batch goes straight to
// its sorted position env_id*T + (rows this env has contributed so far).  16 rows per workgroup; rows with mask == 0
// are skipped, rows of an env that already holds T rows are dropped and counted.  env_ids must be unique within a call.
__global__ void __launch_bounds__(256) store_rows_kernel(pfa_experience ex, int rows, int num_slots, int dp, const float *obs,
                                                        const float *rewards, const uint8_t *dones, const long long *actions,
                                                        const float *logprob, const float *value, const int *env_ids,
                                                        const uint8_t *mask, int *counters, int *stored_dropped) {
    __shared__ int s_t[16], s_slot[16];
    const int r0 = blockIdx.x * 16;
    if (threadIdx.x < 16) {
        const int i = r0 + threadIdx.x;
        int t = -1, slot = 0;
        if (i < rows && (!mask || mask[i])) {
            slot = env_ids ? env_ids[i] : i;
            if (slot >= 0 && slot < num_slots) {
                t = counters[slot];
                if (t >= ex.horizon_T) {
                    t = -1;
                    atomicAdd(stored_dropped + 1, 1);
                }
            } else {
                atomicAdd(stored_dropped + 1, 1);
            }
        }
        s_t[threadIdx.x] = t;
        s_slot[threadIdx.x] = slot;
    }
    __syncthreads();
    const int V = dp / 4;
    for (int idx = threadIdx.x; idx < 16 * V; idx += 256) {
        const int rl = idx / V, c4 = idx - rl * V;
        const int t = s_t[rl];
        if (t < 0) continue;
        *reinterpret_cast<float4 *>(ex.obs + ((size_t)s_slot[rl] * ex.horizon_T + t) * dp + 4 * c4) =
            *reinterpret_cast<const float4 *>(obs + (size_t)(r0 + rl) * dp + 4 * c4);
    }
    if (threadIdx.x < 16 && s_t[threadIdx.x] >= 0) {
        const int i = r0 + threadIdx.x, slot = s_slot[threadIdx.x], t = s_t[threadIdx.x];
        const size_t row = (size_t)slot * ex.horizon_T + t;
        ex.rewards[row] = rewards[i];
        ex.dones[row] = dones[i] ? 1.0f : 0.0f;
        ex.actions[row] = (int)actions[i];
        ex.logprobs[row] = logprob[i];
        ex.values[row] = value[i];
        counters[slot] = t + 1;
        atomicAdd(stored_dropped, 1);
    }
}

// Rows of minibatch `mb` in TIME-MAJOR order: q = t*R + k  ->  flat env-major row (mb + k*nmb)*Th + t
struct TimeMajorMap {
    int mb, nmb, horizon;
    long long R;
    __device__ __forceinline__ long long flat(long long q) const {
        const long long t = q / R, k = q - t * R;
        return ((long long)mb + k * nmb) * horizon + t;
    }
};

__global__ void __launch_bounds__(256) gather_obs_tm_kernel(const float *obs, TimeMajorMap map, long long rows, int dp,
                                                           float *out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int V = dp / 4;
    if (i >= rows * V) return;
    const long long q = i / V;
    const int c4 = (int)(i - q * V);
    *reinterpret_cast<float4 *>(out + q * dp + 4 * c4) = *reinterpret_cast<const float4 *>(obs + map.flat(q) * dp + 4 * c4);
}

// decode_actions + PPO loss (clean_pufferl.py:202-238) on the hidden states of one minibatch (time-major rows) and the
// gradients w.r.t. the 16 padded head outputs and w.r.t. h.  One wavefront per 16-row tile, both small products on the matrix
// cores (v_mfma_f32_16x16x4_f32), the head weights as MFMA B fragments in registers for the whole launch:
//   out[row][o]  = h[row][:] . W2v[o][:] + b2v[o]        32 MFMAs (two accumulator chains); lane (c, g) then holds output o = c of
//                                                        rows 4g .. 4g+3 — the 16 lanes of a DPP row are the 16 outputs of one row,
//                                                        which is the shape the shared log-softmax / arg-max pieces (sampler.hpp) take
//   dh[row][u]   = sum_o dout[row][o] W2v[o][u]          32 MFMAs (8 independent column tiles), dout handed from the accumulator
//                                                        layout to the A-operand layout through a 1 KB wave-private LDS tile
// (the first version ran both as per-lane fma loops over LDS: 75 us per 131 072-row minibatch, LDS-issue bound; now 51 us, of
// which ~35 are the 142 MB of h / dh / dout at HBM rate.  Measured and not taken: dh through an LDS transpose for whole-row stores
// (68 us), tiles of 16 time steps of one segment so that the five experience gathers of a tile share a cache line (55 us: the h
// rows of a tile are then 4 MB apart).)
// stats_partial: [gridDim.x][24].
__global__ void __launch_bounds__(256) lstm_heads_loss_kernel(const float *h, long long rows, TimeMajorMap map,
                                                             pfa_experience ex, const float *params, int dp, int a,
                                                             uint32_t heads, pfa_ppo_hparams hp, const double *adv_stats,
                                                             double global_rows,
                                                             float *dout /* [rows][16] */, float *dh /* [rows][128] */,
                                                             float *stats_partial) {
    constexpr int WS = kLstmH + 4;     // W2v row stride in LDS: 16-byte aligned rows, 16 rows x float4 = all 64 banks once
    __shared__ __attribute__((aligned(16))) float w2v[kOut * WS];
    __shared__ float b2v[kOut];
    __shared__ float dt[4][16][17];   // per wave: dout of its tile as [row][o]
    __shared__ float st[16][8];
    __shared__ float sd[16][16];
    {
        const MlpOffsets off = mlp_offsets(dp, a);
        for (int i = threadIdx.x; i < kOut * kLstmH; i += 256) {
            const int o = i / kLstmH, u = i - o * kLstmH;
            w2v[o * WS + u] = w2v_at(params, off, a, o, u);
        }
        if (threadIdx.x < kOut) b2v[threadIdx.x] = b2v_at(params, off, a, threadIdx.x);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = wave_id(), c = lane & 15, g = lane >> 4;
    const int le = wv * 4 + g, lo = c;      // (le: one of the workgroup's 16 DPP rows; lo: the head output this lane owns)
    // B fragments come from the LDS copy at their point of use (in registers they cost 64 VGPRs and half the occupancy that hides
    // this kernel's gather latencies).  Heads: contraction index k = 16 j + 4 g + e (the A operand is read as float4
    // h[row][16 j + 4 g ..]); dh: contraction index o = 4 kk + g, column tile n.
    const float *wfp = w2v + c * WS + 4 * g, *wbp = w2v + g * WS + c;
    const float bias = b2v[c];
    float adv_mean = 0.0f, adv_den = 1.0f;
    if (hp.norm_adv) {
        const double s1 = adv_stats[2 * map.mb], s2 = adv_stats[2 * map.mb + 1];
        const double mean = s1 / global_rows;
        double var = (s2 - s1 * mean) / (global_rows - 1.0);
        var = var > 0.0 ? var : 0.0;
        adv_mean = (float)mean;
        adv_den = (float)sqrt(var) + 1e-8f;
    }
    const float inv_rows = (float)(1.0 / global_rows);
    const float lo_c = 1.0f - hp.clip_coef, hi_c = 1.0f + hp.clip_coef;
    const bool is_logit = lo < a;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dsum = 0.0f;  // this lane's column of sum_rows dout (the head bias gradient)
    const long long tiles = (rows + 15) / 16;
    for (long long tile = (long long)blockIdx.x * 4 + wv; tile < tiles; tile += (long long)gridDim.x * 4) {
        const long long r0 = tile * 16;
        auto rowq = [&](int i) -> long long { return r0 + i < rows ? r0 + i : -1; };     // row i of the tile, or -1 past the end
        const long long arow = r0 + c < rows ? r0 + c : rows - 1;     // (rows past the end repeat the last one; nothing of them is kept)
        float4 ha[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ha[j] = *reinterpret_cast<const float4 *>(h + arow * kLstmH + 16 * j + 4 * g);
        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const float4 w0 = *reinterpret_cast<const float4 *>(wfp + 16 * j), w1 = *reinterpret_cast<const float4 *>(wfp + 16 * (j + 1));
            o0 = mfma16(ha[j].x, w0.x, o0);
            o1 = mfma16(ha[j + 1].x, w1.x, o1);
            o0 = mfma16(ha[j].y, w0.y, o0);
            o1 = mfma16(ha[j + 1].y, w1.y, o1);
            o0 = mfma16(ha[j].z, w0.z, o0);
            o1 = mfma16(ha[j + 1].z, w1.z, o1);
            o0 = mfma16(ha[j].w, w0.w, o0);
            o1 = mfma16(ha[j + 1].w, w1.w, o1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long row = rowq(4 * g + r);
            const bool ok = row >= 0;
            const long long fr = ok ? map.flat(row) : 0;
            const float w = ok ? 1.0f : 0.0f;
            const int action = ex.actions[fr];
            const float old_logprob = ex.logprobs[fr], old_value = ex.values[fr], adv_raw = ex.advantages[fr], ret = ex.returns[fr];
            const float mine = (o0[r] + o1[r]) + bias;
            // log-softmax over the row's logits
            bool chosen;
            float nl, p, ent, hent, new_logprob;
            if (heads == 0) {
                const float mx = row16_max(is_logit ? mine : -INFINITY);
                const float ev = is_logit ? expf(mine - mx) : 0.0f;
                const float se = row16_sum(ev);
                const float lse = mx + logf(se);
                nl = mine - lse;
                p = ev / se;
                ent = hent = row16_sum(is_logit ? -nl * p : 0.0f);
                chosen = lo == action;
                new_logprob = row16_sum(chosen ? nl : 0.0f);
            } else {  // MultiDiscrete: per-head softmax, sums over heads (sampler.hpp)
                const Row16Eval ev = eval_row16_heads(mine, lo, a, heads, action);
                nl = ev.nl, p = ev.p, hent = ev.head_entropy, ent = ev.entropy, chosen = ev.chosen, new_logprob = ev.logprob;
            }
            const float new_value = row16_sum(lo == a ? mine : 0.0f);
            const float logratio = new_logprob - old_logprob;
            const float ratio = expf(logratio);
            const float adv = hp.norm_adv ? (adv_raw - adv_mean) / adv_den : adv_raw;
            const float pg1 = -adv * ratio, pg2 = -adv * fminf(fmaxf(ratio, lo_c), hi_c);
            const bool inside = ratio >= lo_c && ratio <= hi_c;
            float dpg;  // d pg / d ratio: torch.max tie rule + clamp pass-through (see csrc/ppo_update.hip)
            if (pg1 > pg2) dpg = -adv;
            else if (pg1 < pg2) dpg = inside ? -adv : 0.0f;
            else dpg = inside ? -adv : -0.5f * adv;
            const float scale = inv_rows * w;
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
            float d = 0.0f;
            if (is_logit) d = g_lp * ((chosen ? 1.0f : 0.0f) - p) + hp.ent_coef * scale * p * (nl + hent);
            else if (lo == a) d = dv;
            if (ok) {
                dout[row * kOut + lo] = d;
                dsum += d;
            }
            dt[wv][4 * g + r][lo] = ok ? d : 0.0f;
            if (lo == 0) {
                acc[0] += fmaxf(pg1, pg2) * w;
                acc[1] += v_loss * w;
                acc[2] += ent * w;
                acc[3] += -logratio * w;
                acc[4] += ((ratio - 1.0f) - logratio) * w;
                acc[5] += (fabsf(ratio - 1.0f) > hp.clip_coef ? 1.0f : 0.0f) * w;
            }
        }
        // d loss / d h = dout W2v: A operand lane (c, g) = dout[row c][o = 4 kk + g] from this wave's own LDS tile (a wave's LDS
        // operations complete in order: no barrier between its writes above and these reads)
        __builtin_amdgcn_wave_barrier();
        float da[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) da[kk] = dt[wv][c][4 * kk + g];
        f32x4 dacc[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) dacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int n = 0; n < 8; ++n) dacc[n] = mfma16(da[kk], wbp[4 * kk * WS + 16 * n], dacc[n]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long row = rowq(4 * g + r);
            if (row >= 0) {
#pragma unroll
                for (int n = 0; n < 8; ++n) dh[row * kLstmH + 16 * n + c] = dacc[n][r];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lo == 0)
        for (int i = 0; i < 8; ++i) st[le][i] = i < 6 ? acc[i] : 0.0f;
    sd[le][lo] = dsum;
    __syncthreads();
    if (threadIdx.x < 8) {
        float s = 0.0f;
        for (int e = 0; e < 16; ++e) s += st[e][threadIdx.x];
        stats_partial[(size_t)blockIdx.x * kHeadStatCols + threadIdx.x] = s;
    } else if (threadIdx.x >= 16 && threadIdx.x < 32) {
        float s = 0.0f;
        for (int e = 0; e < 16; ++e) s += sd[e][threadIdx.x - 16];
        stats_partial[(size_t)blockIdx.x * kHeadStatCols + 8 + (threadIdx.x - 16)] = s;
    }
}
// 1024 threads: 32 strided chains per column (8 loss statistics + 16 head-output column sums), then a fixed-order tree.
// The 8 loss statistics are summed in f64 and leave as (hi, lo) float pairs (the tail of the gradient bucket, ppo_tile.hpp).
__global__ void __launch_bounds__(1024) stats_final_kernel(const float *partial, int nblocks, float *loss_pairs16, float *out16) {
    __shared__ double sh[32][32];
    const int i = threadIdx.x & 31, chain = threadIdx.x >> 5;
    double s = 0.0;
    if (i < kHeadStatCols)
        for (int b = chain; b < nblocks; b += 32) s += (double)partial[(size_t)b * kHeadStatCols + i];
    sh[chain][i] = s;
    __syncthreads();
    if (threadIdx.x < kHeadStatCols) {
        double t = 0.0;
        for (int q = 0; q < 32; ++q) t += sh[q][threadIdx.x];
        if (threadIdx.x < 8) {
            const float hi = (float)t;
            loss_pairs16[2 * threadIdx.x] = hi;
            loss_pairs16[2 * threadIdx.x + 1] = (float)(t - (double)hi);
        } else if (out16) {
            out16[threadIdx.x - 8] = (float)t;
        }
    }
}

// f64 pieces of sum(g^2) for the clip norm of a large flat gradient (consumed by adam_clip_kernel).
__global__ void __launch_bounds__(256) sumsq_partial_kernel(const float *g, long long count, double *partials) {
    __shared__ double sh[4];
    const long long per = (count + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per, hi = lo + per < count ? lo + per : count;
    double s = 0.0;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const double v = (double)g[i];
        s += v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane_id() == 0) sh[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// Scatter of the small gradient pieces into the flat gradient vector, one launch: heads product g16 [16][128] (rows < A
// decoder.weight, row A value_head.weight), head bias sums [16], and d b_hh = d b_ih.
__global__ void __launch_bounds__(256) lstm_finish_grads_kernel(float *grads, int dp, int a, const float *g16, const float *bsum16) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const MlpOffsets off = mlp_offsets(dp, a);
    const LstmOffsets lo = lstm_offsets(dp, a);
    if (idx < a * kLstmH) grads[off.w2 + idx] = g16[idx];
    if (idx < kLstmH) grads[off.wv + idx] = g16[a * kLstmH + idx];
    if (idx < a) grads[off.b2 + idx] = bsum16[idx];
    if (idx == 0) grads[off.bv] = bsum16[a];
    if (idx < kLG) grads[lo.b_hh + idx] = grads[lo.b_ih + idx];
}

static unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace pfa

using namespace pfa;

extern "C" int pfa_store_step(const pfa_experience *exp, int32_t t, int32_t num_envs, int32_t obs_stride, const float *obs,
                              const float *rewards, const uint8_t *terminals, const int64_t *actions, const float *logprob,
                              const float *value, pfa_stream_t stream) {
    PFA_REQUIRE(exp && obs && rewards && terminals && actions && logprob && value, "store_step: null buffer");
    PFA_REQUIRE(t >= 0 && t < exp->horizon_T && num_envs >= 1 && obs_stride % 4 == 0, "store_step: bad arguments");
    hipLaunchKernelGGL(store_step_kernel, dim3(blocks_for((long long)num_envs * (obs_stride / 4))), dim3(256), 0,
                       (hipStream_t)stream, *exp, (int)t, (int)num_envs, (int)obs_stride, obs, rewards, terminals,
                       (const long long *)actions, logprob, value);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_store_rows(const pfa_experience *exp, int32_t rows, int32_t num_slots, int32_t obs_stride, const float *obs,
                              const float *rewards, const uint8_t *dones, const int64_t *actions, const float *logprob,
                              const float *value, const int32_t *env_ids, const uint8_t *mask, int32_t *counters,
                              int32_t *stored_dropped, pfa_stream_t stream) {
    PFA_REQUIRE(exp && obs && rewards && dones && actions && logprob && value && counters && stored_dropped, "store_rows: null buffer");
    PFA_REQUIRE(rows >= 0 && num_slots >= 1 && obs_stride >= 4 && obs_stride % 4 == 0 && exp->horizon_T >= 1, "store_rows: bad arguments");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(store_rows_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, (hipStream_t)stream, *exp, (int)rows,
                       (int)num_slots, (int)obs_stride, obs, rewards, dones, (const long long *)actions, logprob, value, env_ids, mask,
                       counters, stored_dropped);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_gather_obs_time_major(const pfa_experience *exp, int64_t batch_rows, int32_t mb, const pfa_ppo_hparams *hp,
                                         int32_t obs_stride, float *out, pfa_stream_t stream) {
    PFA_REQUIRE(exp && hp && out && exp->obs, "gather_obs: null buffer");
    const long long mbs = batch_rows / hp->num_minibatches;
    TimeMajorMap map{mb, hp->num_minibatches, hp->bptt_horizon, mbs / hp->bptt_horizon};
    hipLaunchKernelGGL(gather_obs_tm_kernel, dim3(blocks_for(mbs * (obs_stride / 4))), dim3(256), 0, (hipStream_t)stream, exp->obs,
                       map, mbs, (int)obs_stride, out);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t pfa_lstm_heads_loss_workspace_bytes(void) { return (size_t)1024 * kHeadStatCols * sizeof(float); }

extern "C" int pfa_lstm_heads_loss(const float *h, const pfa_experience *exp, int64_t batch_rows, int32_t mb, const float *params,
                                   const pfa_mlp_dims *dims, const pfa_ppo_hparams *hp, const double *adv_stats,
                                   int64_t global_mb_rows, float *dout, float *dh, float *loss_sums8, float *head_bias_grad16,
                                   void *workspace, pfa_stream_t stream) {
    PFA_REQUIRE(h && exp && params && dims && hp && dout && dh && loss_sums8 && workspace, "lstm_heads_loss: null buffer");
    PFA_REQUIRE(dims->heads == 0 || heads_count(dims->heads, dims->num_actions) >= 1, "lstm_heads_loss: head sizes 0x%x do not sum to %d",
                dims->heads, dims->num_actions);
    PFA_REQUIRE(!hp->norm_adv || adv_stats, "lstm_heads_loss: norm_adv needs adv_stats");
    const long long mbs = batch_rows / hp->num_minibatches;
    TimeMajorMap map{mb, hp->num_minibatches, hp->bptt_horizon, mbs / hp->bptt_horizon};
    const long long tiles = ((mbs + 15) / 16 + 3) / 4;        // workgroup = four 16-row tiles, one per wavefront
    const unsigned grid = (unsigned)(tiles < 1024 ? tiles : 1024);
    float *partial = (float *)workspace;
    hipLaunchKernelGGL(lstm_heads_loss_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, h, mbs, map, *exp, params,
                       dims->obs_stride, dims->num_actions, dims->heads, *hp, adv_stats, (double)global_mb_rows, dout, dh, partial);
    PFA_LAUNCH_CHECK();
    hipLaunchKernelGGL(stats_final_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, partial, (int)grid, loss_sums8, head_bias_grad16);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_lstm_finish_grads(float *grads, const pfa_mlp_dims *dims, const float *g16, const float *bsum16,
                                     pfa_stream_t stream) {
    PFA_REQUIRE(grads && dims && g16 && bsum16, "lstm_finish_grads: null buffer");
    PFA_REQUIRE(dims->num_actions >= 1 && dims->num_actions <= 15, "lstm_finish_grads: num_actions must be in 1..15");
    hipLaunchKernelGGL(lstm_finish_grads_kernel, dim3(16 * kLstmH / 256), dim3(256), 0, (hipStream_t)stream, grads, dims->obs_stride,
                       dims->num_actions, g16, bsum16);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_sumsq_partials(const float *grads, int64_t count, double *partials, int32_t n, pfa_stream_t stream) {
    PFA_REQUIRE(grads && partials && count >= 1 && n >= 1 && n <= 4096, "sumsq_partials: bad arguments");
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, grads, (long long)count, partials);
    PFA_LAUNCH_CHECK();
    return 0;
}
