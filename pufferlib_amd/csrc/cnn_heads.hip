// cnn_heads.hip — the head side of the NatureCNN policy (pufferlib/models.py:152-157 decode_actions: actor Linear(512, A),
// value_fn Linear(512, 1)) on the 512-wide hidden vector the conv stack + Linear(3136, 512) of csrc/igemm.hip produce:
//   cnn_heads_sample   rollout: logits, value, sample_logits (cleanrl.py:25-47) -> action, log-prob, entropy, value
//   cnn_heads_loss     training: the same heads, the PPO loss of clean_pufferl.py:202-238, d loss / d (head outputs) [rows][16],
//                      d loss / d (pre-ReLU hidden) [rows][512], the six loss sums (f64, (hi, lo) float pairs)
//   cnn_gather_frames  the uint8 frames of a chunk of minibatch rows, copied next to each other (rows of a minibatch are
//                      bptt_horizon-long segments of the env-major experience, clean_pufferl.py:455-457)
// 16 lanes per row (lane lo = head output lo), the row's hidden vector in LDS, head weights in LDS (padded rows: conflict-free),
// the 16-lane log-softmax / arg-max pieces shared with the MLP and LSTM policies (sampler.hpp), so all three sample and score
// with identical arithmetic.  VALU work: 2 x 512 x (A + 1) flop per row against ~56 MFLOP in the conv stack.
#include "common.hpp"
#include "lane_ops.hpp"
#include "mlp_tile.hpp"
#include "ppo_tile.hpp"
#include "sampler.hpp"

namespace pfa {

constexpr int kCnnH = 512;

struct CnnHeads {   // actor.weight [A][512], actor.bias [A], value_fn.weight [1][512], value_fn.bias [1] (torch layout)
    const float *w2, *b2, *wv, *bv;
    int a;
};

__device__ __forceinline__ void cnn_stage_heads(const CnnHeads &hd, float *w2v /* [16][513] */, float *b2v) {
    for (int i = threadIdx.x; i < kOut * kCnnH; i += blockDim.x) {
        const int o = i / kCnnH, u = i - o * kCnnH;
        w2v[o * (kCnnH + 1) + u] = o < hd.a ? hd.w2[o * kCnnH + u] : (o == hd.a ? hd.wv[u] : 0.0f);
    }
    for (int i = threadIdx.x; i < kOut; i += blockDim.x) b2v[i] = i < hd.a ? hd.b2[i] : (i == hd.a ? hd.bv[0] : 0.0f);
}
__device__ __forceinline__ float cnn_head_dot(const float *hrow, const float *w2v, const float *b2v, int lo) {
    float acc = b2v[lo];
    const float *w = w2v + lo * (kCnnH + 1);
#pragma unroll 8
    for (int u = 0; u < kCnnH; ++u) acc = fmaf(hrow[u], w[u], acc);   // k-ordered fma chain, like nn.Linear's fp32 dot
    return acc;
}

__global__ void __launch_bounds__(256) cnn_heads_sample_kernel(const float *h, long long rows, CnnHeads hd, const float *noise, uint64_t seed,
                                                              uint64_t step, long long row_offset, long long *actions, float *logprob,
                                                              float *entropy, float *value) {
    extern __shared__ float lds[];
    float *w2v = lds, *b2v = w2v + kOut * (kCnnH + 1), *hs = b2v + kOut;   // hs [16][512]
    cnn_stage_heads(hd, w2v, b2v);
    const int le = threadIdx.x >> 4, lo = threadIdx.x & 15, a = hd.a;
    const long long tiles = (rows + 15) / 16;
    for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        __syncthreads();
        for (int i = threadIdx.x; i < 16 * kCnnH; i += 256) {
            const long long r = tile * 16 + i / kCnnH;
            hs[i] = r < rows ? h[r * kCnnH + i % kCnnH] : 0.0f;
        }
        __syncthreads();
        const long long row = tile * 16 + le;
        const bool ok = row < rows;
        const float mine = cnn_head_dot(hs + le * kCnnH, w2v, b2v, lo);
        const float q = ok ? noise_lane(noise ? noise + row * a : nullptr, seed, step, (uint64_t)(row_offset + row), lo, a) : 1.0f;
        const LaneSample sm = sample_row16(mine, lo, a, q);
        if (ok && lo == 0) {
            actions[row] = sm.action;
            logprob[row] = sm.logprob;
            if (entropy) entropy[row] = sm.entropy;
            value[row] = sm.value;
        }
    }
}

// rows = a chunk [q0, q0 + rows) of minibatch `map.mb`; ex.* are read at map.flat(q0 + row).
__global__ void __launch_bounds__(256) cnn_heads_loss_kernel(const float *h, long long rows, RowMap map, long long q0, pfa_experience ex, CnnHeads hd,
                                                            pfa_ppo_hparams hp, const double *adv_stats, double global_rows,
                                                            float *dout /* [rows][16] */, float *dh /* [rows][512] */,
                                                            double *stats_partial /* [gridDim.x][8] */) {
    extern __shared__ float lds[];
    float *w2v = lds, *b2v = w2v + kOut * (kCnnH + 1), *hs = b2v + kOut;
    __shared__ double st[16][8];
    cnn_stage_heads(hd, w2v, b2v);
    const int le = threadIdx.x >> 4, lo = threadIdx.x & 15, a = hd.a;
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
    double acc[6] = {0, 0, 0, 0, 0, 0};
    const long long tiles = (rows + 15) / 16;
    for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        __syncthreads();
        for (int i = threadIdx.x; i < 16 * kCnnH; i += 256) {
            const long long r = tile * 16 + i / kCnnH;
            hs[i] = r < rows ? h[r * kCnnH + i % kCnnH] : 0.0f;
        }
        __syncthreads();
        const long long row = tile * 16 + le;
        const bool ok = row < rows;
        const long long fr = ok ? map.flat(q0 + row) : 0;
        const float w = ok ? 1.0f : 0.0f;
        const int action = ex.actions[fr];
        const float old_logprob = ex.logprobs[fr], old_value = ex.values[fr], adv_raw = ex.advantages[fr], ret = ex.returns[fr];
        const float mine = cnn_head_dot(hs + le * kCnnH, w2v, b2v, lo);
        const bool is_logit = lo < a;
        const float mx = row16_max(is_logit ? mine : -INFINITY);
        const float ev = is_logit ? expf(mine - mx) : 0.0f;
        const float se = row16_sum(ev);
        const float lse = mx + logf(se);
        const float nl = mine - lse, p = ev / se;
        const float ent = row16_sum(is_logit ? -nl * p : 0.0f);
        const bool chosen = lo == action;
        const float new_logprob = row16_sum(chosen ? nl : 0.0f);
        const float new_value = row16_sum(lo == a ? mine : 0.0f);
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
        if (is_logit) d = g_lp * ((chosen ? 1.0f : 0.0f) - p) + hp.ent_coef * scale * p * (nl + ent);
        else if (lo == a) d = dv;
        if (ok) dout[row * kOut + lo] = d;
        // d loss / d h[u] = sum_o d_o W2v[o][u]; lane lo owns u = lo, lo + 16, ...
        float dhv[kCnnH / 16];
#pragma unroll
        for (int j = 0; j < kCnnH / 16; ++j) dhv[j] = 0.0f;
        for (int o = 0; o <= a; ++o) {
            const float d_o = __shfl(d, (lane_id() & 48) | o, 64);
#pragma unroll
            for (int j = 0; j < kCnnH / 16; ++j) dhv[j] = fmaf(d_o, w2v[o * (kCnnH + 1) + lo + 16 * j], dhv[j]);
        }
        if (ok) {   // the hidden vector is relu(Linear(...)): hand back d loss / d (pre-activation), i.e. masked by relu'
#pragma unroll
            for (int j = 0; j < kCnnH / 16; ++j) dh[row * kCnnH + lo + 16 * j] = hs[le * kCnnH + lo + 16 * j] > 0.0f ? dhv[j] : 0.0f;
        }
        if (lo == 0) {
            acc[0] += (double)(fmaxf(pg1, pg2) * w);
            acc[1] += (double)(v_loss * w);
            acc[2] += (double)(ent * w);
            acc[3] += (double)(-logratio * w);
            acc[4] += (double)(((ratio - 1.0f) - logratio) * w);
            acc[5] += (double)((fabsf(ratio - 1.0f) > hp.clip_coef ? 1.0f : 0.0f) * w);
        }
    }
    __syncthreads();
    if (lo == 0)
        for (int i = 0; i < 8; ++i) st[le][i] = i < 6 ? acc[i] : 0.0;
    __syncthreads();
    if (threadIdx.x < 8) {
        double s = 0.0;
        for (int r = 0; r < 16; ++r) s += st[r][threadIdx.x];
        stats_partial[(size_t)blockIdx.x * 8 + threadIdx.x] = s;
    }
}

// loss_pairs16 (+)= the (hi, lo) pairs of the chunk's six sums (a minibatch is processed in chunks: accumulate != 0 adds)
__global__ void cnn_stats_final_kernel(const double *partial, int nblocks, float *loss_pairs16, int accumulate) {   // one wave
    const int i = threadIdx.x & 7, part = threadIdx.x >> 3;   // 8 sums x 8 interleaved block ranges
    double s = 0.0;
    for (int b = part; b < nblocks; b += 8) s += partial[(size_t)b * 8 + i];
    s += __shfl_xor(s, 8, 64);      // fixed combination order
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (threadIdx.x >= 8) return;
    if (accumulate) s += (double)loss_pairs16[2 * i] + (double)loss_pairs16[2 * i + 1];
    const float hi = (float)s;
    loss_pairs16[2 * i] = hi;
    loss_pairs16[2 * i + 1] = (float)(s - (double)hi);
}

// frames [B][frame_bytes] (env-major experience) -> out[rows][frame_bytes] for minibatch rows q0 .. q0 + rows - 1
__global__ void __launch_bounds__(256) cnn_gather_frames_kernel(const uint8_t *frames, long long frame_bytes, RowMap map, long long q0, long long rows,
                                                               uint8_t *out) {
    const long long per_row = frame_bytes / 16;   // 16-byte pieces
    const long long total = rows * per_row;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / per_row, c16 = i - r * per_row;
        const long long fr = map.flat(q0 + r);
        reinterpret_cast<uint4 *>(out + r * frame_bytes)[c16] = reinterpret_cast<const uint4 *>(frames + fr * frame_bytes)[c16];
    }
}

static size_t cnn_heads_lds() { return (size_t)(kOut * (kCnnH + 1) + kOut + 16 * kCnnH) * sizeof(float); }

}  // namespace pfa

using namespace pfa;

extern "C" int pfa_cnn_heads_sample(const float *h, int64_t rows, const float *actor_w, const float *actor_b, const float *value_w,
                                    const float *value_b, int32_t num_actions, const float *noise, const pfa_noise_key *key, int64_t row_offset,
                                    int64_t *actions, float *logprob, float *entropy, float *value, pfa_stream_t stream) {
    PFA_REQUIRE(rows >= 0 && h && actor_w && actor_b && value_w && value_b && actions && logprob && value, "cnn.heads_sample: null buffer");
    PFA_REQUIRE(num_actions >= 1 && num_actions <= 15, "cnn.heads_sample: num_actions must be in 1..15");
    PFA_REQUIRE(noise || key, "cnn.heads_sample: need an explicit noise tensor or a Philox key");
    if (rows == 0) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        PFA_CHECK_HIP(hipFuncSetAttribute((const void *)cnn_heads_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cnn_heads_lds()));
        attr_set = true;
    }
    const long long tiles = (rows + 15) / 16;
    const unsigned grid = (unsigned)(tiles < 1024 ? tiles : 1024);
    CnnHeads hd{actor_w, actor_b, value_w, value_b, (int)num_actions};
    ScopedKernelTimer timer("cnn_heads_sample", (hipStream_t)stream);
    hipLaunchKernelGGL(cnn_heads_sample_kernel, dim3(grid), dim3(256), cnn_heads_lds(), (hipStream_t)stream, h, (long long)rows, hd, noise,
                       key ? key->seed : 0, key ? key->step : 0, (long long)row_offset, (long long *)actions, logprob, entropy, value);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t pfa_cnn_heads_loss_workspace_bytes(void) { return (size_t)1024 * 8 * sizeof(double); }

extern "C" int pfa_cnn_heads_loss(const float *h, const pfa_experience *exp, int64_t batch_rows, int32_t mb, int64_t q0, int64_t rows,
                                  const float *actor_w, const float *actor_b, const float *value_w, const float *value_b, int32_t num_actions,
                                  const pfa_ppo_hparams *hp, const double *adv_stats, int64_t global_mb_rows, float *dout, float *dh,
                                  float *loss_pairs16, int32_t accumulate, void *workspace, pfa_stream_t stream) {
    PFA_REQUIRE(h && exp && hp && dout && dh && loss_pairs16 && workspace && actor_w && actor_b && value_w && value_b, "cnn.heads_loss: null buffer");
    PFA_REQUIRE(num_actions >= 1 && num_actions <= 15, "cnn.heads_loss: num_actions must be in 1..15");
    PFA_REQUIRE(hp->num_minibatches >= 1 && batch_rows % hp->num_minibatches == 0 && mb >= 0 && mb < hp->num_minibatches, "cnn.heads_loss: bad minibatch");
    const int64_t mbs = batch_rows / hp->num_minibatches;
    PFA_REQUIRE(q0 >= 0 && rows >= 1 && q0 + rows <= mbs, "cnn.heads_loss: chunk outside the minibatch");
    PFA_REQUIRE(!hp->norm_adv || adv_stats, "cnn.heads_loss: norm_adv needs adv_stats");
    static bool attr_set = false;
    if (!attr_set) {
        PFA_CHECK_HIP(hipFuncSetAttribute((const void *)cnn_heads_loss_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cnn_heads_lds()));
        attr_set = true;
    }
    const long long tiles = (rows + 15) / 16;
    const unsigned grid = (unsigned)(tiles < 1024 ? tiles : 1024);
    CnnHeads hd{actor_w, actor_b, value_w, value_b, (int)num_actions};
    RowMap map{mb, hp->num_minibatches, hp->bptt_horizon};
    ScopedKernelTimer timer("cnn_heads_loss", (hipStream_t)stream);
    hipLaunchKernelGGL(cnn_heads_loss_kernel, dim3(grid), dim3(256), cnn_heads_lds(), (hipStream_t)stream, h, (long long)rows, map, (long long)q0, *exp,
                       hd, *hp, adv_stats, (double)global_mb_rows, dout, dh, (double *)workspace);
    PFA_LAUNCH_CHECK();
    hipLaunchKernelGGL(cnn_stats_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double *)workspace, (int)grid, loss_pairs16, (int)accumulate);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_cnn_gather_frames(const uint8_t *frames, int64_t frame_bytes, int64_t batch_rows, int32_t mb, const pfa_ppo_hparams *hp,
                                     int64_t q0, int64_t rows, uint8_t *out, pfa_stream_t stream) {
    PFA_REQUIRE(frames && out && hp && frame_bytes >= 16 && frame_bytes % 16 == 0, "cnn.gather_frames: frames must be a multiple of 16 bytes");
    PFA_REQUIRE(hp->num_minibatches >= 1 && batch_rows % hp->num_minibatches == 0 && mb >= 0 && mb < hp->num_minibatches, "cnn.gather_frames: bad minibatch");
    PFA_REQUIRE(q0 >= 0 && rows >= 1 && q0 + rows <= batch_rows / hp->num_minibatches, "cnn.gather_frames: chunk outside the minibatch");
    RowMap map{mb, hp->num_minibatches, hp->bptt_horizon};
    const long long total = rows * (frame_bytes / 16);
    const unsigned grid = (unsigned)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
    hipLaunchKernelGGL(cnn_gather_frames_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, frames, (long long)frame_bytes, map, (long long)q0,
                       (long long)rows, out);
    PFA_LAUNCH_CHECK();
    return 0;
}
