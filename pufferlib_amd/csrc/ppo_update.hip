// ppo_update.hip — the minibatch loop of clean_pufferl.train (clean_pufferl.py:175-258) for the MLP policy:
//   forward (models.py:41-62) -> sample_logits with given actions (cleanrl.py:25-47) -> PPO loss (:202-238)
//   -> backward -> clip_grad_norm_ + Adam (:240-244).
//
// Kernel A  ppo_mlp_grad_kernel   fused forward + loss + backward over 16-row tiles, one tile per wavefront at a
//           time, fp32 MFMA (v_mfma_f32_16x16x4_f32) for every contraction; the hidden layer never leaves
//           registers/LDS.  Each workgroup emits one partial flat gradient (+6 loss sums).  MFMA-bound:
//           352 MFMA x 32 cycles per 16 rows vs 16*obs_stride*4 B of HBM reads.
// Kernel B  ppo_reduce_kernel     deterministic sum of the workgroup partials -> flat gradient.
// Kernel C  adam_clip_kernel      global grad-norm clip + Adam, single workgroup (P ~ 1e4).
// Advantage normalisation statistics come from adv_stats kernels (fp64 sums, once per update, because the
// minibatch partition is fixed: no shuffle, clean_pufferl.py:455-457).
//
// Layout bookkeeping of kernel A (c = lane&15, g = lane>>4; see mlp_tile.hpp for the MFMA conventions):
//   hidden^T, out^T, dout^T  : C fragments with rows = features, cols = batch rows (lane c <-> row)
//   dh[row][u] = dout . W2v  : uses dout^T's C fragment as the A operand (A = C^T), giving C rows = batch rows,
//                              which is what the two weight-gradient contractions over batch rows need as B operand
//   dW1^T[k][u] += X[row][k]^T . dh[row][u]      A from the LDS X tile
//   dW2v^T[u][o] += hidden[row][u]^T . dout[row][o]   A from the LDS hidden tile, B from the LDS dout tile
#include "common.hpp"
#include "mlp_tile.hpp"

namespace pfa {

constexpr int kGradThreads = 256;
constexpr int kGradWaves = kGradThreads / 64;
constexpr int kNumStats = 8;  // 6 used: pg, v, entropy, old_kl, kl, clipfrac

template <int DP>
struct GradLds {
    static constexpr int XS = XTile<DP>::XS;
    static constexpr int HS = kHidden + 4;  // hidden tile row stride (16B aligned rows, conflict-free reads)
    static constexpr int DS = 20;           // dout tile row stride
    static constexpr int kWaveFloats = 16 * XS + 16 * HS + 16 * DS;
    static constexpr int kTableFloats = 3 * kMT * 4 * 64;  // b1, w2 (A frags), w2b (B frags), lane-major
    static constexpr int kFloats = kGradWaves * kWaveFloats + kTableFloats;
};

struct RowMap {  // minibatch row q -> flat env-major experience row (clean_pufferl.py:455-457)
    int mb, nmb, horizon;
    __device__ __forceinline__ long long flat(long long q) const {
        const long long k = q / horizon, h = q - k * horizon;
        return ((long long)mb + k * nmb) * horizon + h;
    }
};

template <int DP>
__global__ void __launch_bounds__(kGradThreads, 1)
    ppo_mlp_grad_kernel(pfa_experience ex, RowMap map, long long mb_rows, const float *params, int a, pfa_ppo_hparams hp,
                        const double *adv_stats /* [nmb][2] */, double global_rows, float *partials) {
    using L = GradLds<DP>;
    constexpr int XS = L::XS, HS = L::HS, DS = L::DS, KT = DP / 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = lane_id(), wv = wave_id(), c = lane & 15, g = lane >> 4;
    float *xs = lds + wv * L::kWaveFloats;
    float *hs = xs + 16 * XS;
    float *ds = hs + 16 * HS;
    float *tab_b1 = lds + kGradWaves * L::kWaveFloats;
    float *tab_w2 = tab_b1 + kMT * 4 * 64;
    float *tab_w2b = tab_w2 + kMT * 4 * 64;
    const MlpOffsets off = mlp_offsets(DP, a);

    // fragment tables shared by the 4 waves (lane-major so every read is lds[idx*64 + lane])
    for (int i = threadIdx.x; i < kMT * 4 * 64; i += kGradThreads) {
        const int ln = i & 63, idx = i >> 6, m = idx >> 2, r = idx & 3, cc = ln & 15, gg = ln >> 4;
        tab_b1[i] = params[off.b1 + 16 * m + 4 * gg + r];
        tab_w2[i] = w2v_at(params, off, a, cc, 16 * m + 4 * gg + r);   // A[i=o=cc][k-slot gg] for u = 16m+4gg+r
        tab_w2b[i] = w2v_at(params, off, a, 4 * gg + r, 16 * m + cc);   // B[k-slot gg][j=u=16m+cc] for o = 4gg+r
    }
    float w1f[kMT][DP / 4];
#pragma unroll
    for (int m = 0; m < kMT; ++m)
#pragma unroll
        for (int kk = 0; kk < DP / 4; ++kk) w1f[m][kk] = params[off.w1 + (16 * m + c) * DP + 4 * kk + g];
    float bo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bo[r] = b2v_at(params, off, a, 4 * g + r);

    // advantage normalisation (clean_pufferl.py:211-213): unbiased std over the GLOBAL minibatch
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

    f32x4 acc_dw1[KT][kMT];
    f32x4 acc_dw2[kMT];
    float db1[kMT], db2[4], stats[6];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int m = 0; m < kMT; ++m) acc_dw1[kt][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < kMT; ++m) {
        acc_dw2[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        db1[m] = 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) db2[r] = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) stats[i] = 0.0f;
    __syncthreads();

    const long long tiles = mb_rows / 16;
    const long long wave_global = (long long)blockIdx.x * kGradWaves + wv;
    const long long wave_count = (long long)gridDim.x * kGradWaves;
    for (long long tile = wave_global; tile < tiles; tile += wave_count) {
        // ---- stage X tile (16 rows of DP floats) ------------------------------------------------
        {
            constexpr int V = DP / 4;
#pragma unroll
            for (int j = 0; j < (16 * V + 63) / 64; ++j) {
                const int idx = lane + 64 * j;
                if (idx < 16 * V) {
                    const int r = idx / V, c4 = idx - r * V;
                    const long long fr = map.flat(tile * 16 + r);
                    const float4 v = *reinterpret_cast<const float4 *>(ex.obs + fr * DP + 4 * c4);
                    float2 *d = reinterpret_cast<float2 *>(xs + r * XS + 4 * c4);
                    d[0] = make_float2(v.x, v.y);
                    d[1] = make_float2(v.z, v.w);
                }
            }
        }
        // per-row scalars (lane c <-> row c of the tile, replicated over the 4 lane groups)
        const long long frow = map.flat(tile * 16 + c);
        const int action = ex.actions[frow];
        const float old_logprob = ex.logprobs[frow];
        const float old_value = ex.values[frow];
        const float adv_raw = ex.advantages[frow];
        const float ret = ex.returns[frow];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();

        // ---- forward: hidden^T then out^T ----------------------------------------------------------
        f32x4 h[kMT];
#pragma unroll
        for (int m = 0; m < kMT; ++m)
            h[m] = f32x4{tab_b1[(4 * m + 0) * 64 + lane], tab_b1[(4 * m + 1) * 64 + lane], tab_b1[(4 * m + 2) * 64 + lane],
                         tab_b1[(4 * m + 3) * 64 + lane]};
#pragma unroll
        for (int kk = 0; kk < DP / 4; ++kk) {
            const float b = xs[c * XS + 4 * kk + g];
#pragma unroll
            for (int m = 0; m < kMT; ++m) h[m] = mfma16(w1f[m][kk], b, h[m]);
        }
        f32x4 o0 = f32x4{bo[0], bo[1], bo[2], bo[3]}, o1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < kMT; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) h[m][r] = fmaxf(h[m][r], 0.0f);
            // hidden tile, row-major [row][u], for the relu mask and the dW2v contraction
            *reinterpret_cast<f32x4 *>(hs + c * HS + 16 * m + 4 * g) = h[m];
        }
#pragma unroll
        for (int m = 0; m < kMT; m += 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o0 = mfma16(tab_w2[(4 * m + r) * 64 + lane], h[m][r], o0);
                o1 = mfma16(tab_w2[(4 * (m + 1) + r) * 64 + lane], h[m + 1][r], o1);
            }
        }
        const f32x4 out = o0 + o1;  // out^T[o = 4g + r][row = c]

        // ---- loss and d(loss)/d(out) -----------------------------------------------------------------
        // log-softmax over the A logits of row c, spread over lane groups: reduce with xor 16 / 32.
        float lmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * g + r < a) lmax = fmaxf(lmax, out[r]);
        lmax = fmaxf(lmax, __shfl_xor(lmax, 16, 64));
        lmax = fmaxf(lmax, __shfl_xor(lmax, 32, 64));
        float se = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * g + r < a) se += expf(out[r] - lmax);
        se += __shfl_xor(se, 16, 64);
        se += __shfl_xor(se, 32, 64);
        const float lse = lmax + logf(se);
        float nl[4], p[4], ent = 0.0f, new_logprob = 0.0f, new_value = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 4 * g + r;
            nl[r] = out[r] - lse;
            p[r] = o < a ? expf(nl[r]) : 0.0f;
            if (o < a) ent -= nl[r] * p[r];
            if (o == action) new_logprob = nl[r];
            if (o == a) new_value = out[r];
        }
        ent += __shfl_xor(ent, 16, 64);
        ent += __shfl_xor(ent, 32, 64);
        new_logprob += __shfl_xor(new_logprob, 16, 64);
        new_logprob += __shfl_xor(new_logprob, 32, 64);
        new_value += __shfl_xor(new_value, 16, 64);
        new_value += __shfl_xor(new_value, 32, 64);

        const float logratio = new_logprob - old_logprob;
        const float ratio = expf(logratio);
        const float adv = hp.norm_adv ? (adv_raw - adv_mean) / adv_den : adv_raw;
        const float lo = 1.0f - hp.clip_coef, hi = 1.0f + hp.clip_coef;
        const float pg1 = -adv * ratio;
        const float pg2 = -adv * fminf(fmaxf(ratio, lo), hi);
        const float pg = fmaxf(pg1, pg2);
        const bool inside = ratio >= lo && ratio <= hi;
        // d pg / d ratio under torch.max tie rule (grad/2 to each side) and clamp's pass-through inside [lo, hi]
        float dpg_dratio;
        if (pg1 > pg2) dpg_dratio = -adv;
        else if (pg1 < pg2) dpg_dratio = inside ? -adv : 0.0f;
        else dpg_dratio = inside ? -adv : -0.5f * adv;
        const float g_lp = dpg_dratio * ratio * inv_rows;  // d loss / d new_logprob

        float v_loss, dv;
        if (hp.clip_vloss) {
            const float du = new_value - ret;
            const float vl_u = du * du;
            const float delta = new_value - old_value;
            const float v_clipped = old_value + fminf(fmaxf(delta, -hp.vf_clip_coef), hp.vf_clip_coef);
            const float dc = v_clipped - ret;
            const float vl_c = dc * dc;
            const bool vin = delta >= -hp.vf_clip_coef && delta <= hp.vf_clip_coef;
            v_loss = 0.5f * fmaxf(vl_u, vl_c);
            const float gu = 2.0f * du, gc = vin ? 2.0f * dc : 0.0f;
            const float sel = vl_u > vl_c ? gu : (vl_u < vl_c ? gc : 0.5f * (gu + gc));
            dv = 0.5f * sel;
        } else {
            const float du = new_value - ret;
            v_loss = 0.5f * du * du;
            dv = du;
        }
        dv *= hp.vf_coef * inv_rows;

        f32x4 dout;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 4 * g + r;
            float d = 0.0f;
            if (o < a) {
                // d new_logprob/d logit_o = [o==action] - p_o ; d entropy/d logit_o = -p_o (nl_o + H)
                d = g_lp * ((o == action ? 1.0f : 0.0f) - p[r]) + hp.ent_coef * inv_rows * p[r] * (nl[r] + ent);
            } else if (o == a) {
                d = dv;
            }
            dout[r] = d;
            db2[r] += d;
        }
        if (g == 0) {  // one lane group owns the per-row scalars
            stats[0] += pg;
            stats[1] += v_loss;
            stats[2] += ent;
            stats[3] += -logratio;
            stats[4] += (ratio - 1.0f) - logratio;
            stats[5] += fabsf(ratio - 1.0f) > hp.clip_coef ? 1.0f : 0.0f;
        }
        *reinterpret_cast<f32x4 *>(ds + c * DS + 4 * g) = dout;  // dout[row=c][o=4g..4g+3]
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();

        // ---- backward -----------------------------------------------------------------------------------
        // B fragments of dout[row][o] (k-slot g <-> row 4g+r, j = o = c) for the dW2v contraction
        float dfrag[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) dfrag[r] = ds[(4 * g + r) * DS + c];
        // A fragments of X^T (i = k = 16kt + c, k-slot g <-> row 4g+r) for the dW1 contraction
        float xa[KT][4];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) xa[kt][r] = xs[(4 * g + r) * XS + 16 * kt + c];

#pragma unroll
        for (int m = 0; m < kMT; ++m) {
            // dh[row = 4g+r][u = 16m + c] = sum_o dout[row][o] W2v[o][u]; A = dout^T C-fragment (A = C^T)
            f32x4 dh = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) dh = mfma16(dout[r], tab_w2b[(4 * m + r) * 64 + lane], dh);
            float hrow[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                hrow[r] = hs[(4 * g + r) * HS + 16 * m + c];  // hidden[row = 4g+r][u = 16m+c]
                dh[r] = hrow[r] > 0.0f ? dh[r] : 0.0f;       // relu'
                db1[m] += dh[r];
            }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc_dw1[kt][m] = mfma16(xa[kt][r], dh[r], acc_dw1[kt][m]);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc_dw2[m] = mfma16(hrow[r], dfrag[r], acc_dw2[m]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }

    // ---- reduce the 4 waves' accumulators into one workgroup partial, in fixed wave order -------------------
    __syncthreads();
    float *red = lds;  // reuse the tile regions (tables are no longer needed either)
    const int PP = off.count + kNumStats;
#pragma unroll
    for (int m = 0; m < kMT; ++m) {
        db1[m] += __shfl_xor(db1[m], 16, 64);
        db1[m] += __shfl_xor(db1[m], 32, 64);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 1; s < 16; s <<= 1) db2[r] += __shfl_xor(db2[r], s, 64);
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int s = 1; s < 16; s <<= 1) stats[i] += __shfl_xor(stats[i], s, 64);

    for (int turn = 0; turn < kGradWaves; ++turn) {
        if (wv == turn) {
            const bool first = turn == 0;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int m = 0; m < kMT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = off.w1 + (16 * m + c) * DP + 16 * kt + 4 * g + r;
                        red[idx] = (first ? 0.0f : red[idx]) + acc_dw1[kt][m][r];
                    }
#pragma unroll
            for (int m = 0; m < kMT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int u = 16 * m + 4 * g + r;
                    if (c <= a) {
                        const int idx = c < a ? off.w2 + c * kHidden + u : off.wv + u;
                        red[idx] = (first ? 0.0f : red[idx]) + acc_dw2[m][r];
                    }
                }
            if (g == 0) {
#pragma unroll
                for (int m = 0; m < kMT; ++m) {
                    const int idx = off.b1 + 16 * m + c;
                    red[idx] = (first ? 0.0f : red[idx]) + db1[m];
                }
            }
            if (c == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = 4 * g + r;
                    if (o <= a) {
                        const int idx = o < a ? off.b2 + o : off.bv;
                        red[idx] = (first ? 0.0f : red[idx]) + db2[r];
                    }
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < kNumStats; ++i) {
                    const int idx = off.count + i;
                    red[idx] = (first ? 0.0f : red[idx]) + (i < 6 ? stats[i] : 0.0f);
                }
            }
        }
        __syncthreads();
    }
    float *dst = partials + (size_t)blockIdx.x * PP;
    for (int i = threadIdx.x; i < PP; i += kGradThreads) dst[i] = red[i];
}

// Sum of workgroup partials in a fixed order.  Block = 32 params x 8 slices.
__global__ void __launch_bounds__(256) ppo_reduce_kernel(const float *partials, int nparts, int pp, float *grads) {
    __shared__ float sh[8][33];
    const int pl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int p = blockIdx.x * 32 + pl;
    float acc = 0.0f;
    if (p < pp)
        for (int i = sl; i < nparts; i += 8) acc += partials[(size_t)i * pp + p];
    sh[sl][pl] = acc;
    __syncthreads();
    if (sl == 0 && p < pp) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += sh[i][pl];
        grads[p] = s;
    }
}

// Per-minibatch advantage sums (f64).  grid = (chunks, nmb); deterministic two-stage reduction.
constexpr int kAdvChunks = 64;
__global__ void __launch_bounds__(256) adv_stats_partial_kernel(const float *adv, RowMap base, long long mb_rows,
                                                               double *partial /* [nmb][kAdvChunks][2] */) {
    __shared__ double sh1[256], sh2[256];
    RowMap map = base;
    map.mb = blockIdx.y;
    const long long per = (mb_rows + kAdvChunks - 1) / kAdvChunks;
    const long long lo = (long long)blockIdx.x * per, hi = lo + per < mb_rows ? lo + per : mb_rows;
    double s1 = 0.0, s2 = 0.0;
    for (long long q = lo + threadIdx.x; q < hi; q += 256) {
        const double v = (double)adv[map.flat(q)];
        s1 += v;
        s2 += v * v;
    }
    sh1[threadIdx.x] = s1;
    sh2[threadIdx.x] = s2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sh1[threadIdx.x] += sh1[threadIdx.x + s];
            sh2[threadIdx.x] += sh2[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[((size_t)blockIdx.y * kAdvChunks + blockIdx.x) * 2 + 0] = sh1[0];
        partial[((size_t)blockIdx.y * kAdvChunks + blockIdx.x) * 2 + 1] = sh2[0];
    }
}
__global__ void adv_stats_final_kernel(const double *partial, int nmb, double *stats) {
    const int mb = blockIdx.x * blockDim.x + threadIdx.x;
    if (mb >= nmb) return;
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < kAdvChunks; ++i) {
        s1 += partial[((size_t)mb * kAdvChunks + i) * 2 + 0];
        s2 += partial[((size_t)mb * kAdvChunks + i) * 2 + 1];
    }
    stats[2 * mb] = s1;
    stats[2 * mb + 1] = s2;
}

// clip_grad_norm_ + torch.optim.Adam (single-tensor path) on the flat parameter vector.
constexpr int kAdamThreads = 1024;
__global__ void __launch_bounds__(kAdamThreads) adam_clip_kernel(float *params, const float *grads, float *exp_avg,
                                                                float *exp_avg_sq, long long count, float lr, float beta1,
                                                                float beta2, float eps, long long step, float max_grad_norm,
                                                                float grad_scale, const float *loss_sums, float *losses,
                                                                float loss_scale) {
    __shared__ double sh[kAdamThreads];
    double ss = 0.0;
    for (long long i = threadIdx.x; i < count; i += kAdamThreads) {
        const float gi = grads[i] * grad_scale;
        ss += (double)gi * (double)gi;
    }
    sh[threadIdx.x] = ss;
    __syncthreads();
    for (int s = kAdamThreads / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    const float total_norm = (float)sqrt(sh[0]);
    float clip = 1.0f;
    if (max_grad_norm > 0.0f) {
        clip = max_grad_norm / (total_norm + 1e-6f);
        clip = clip > 1.0f ? 1.0f : clip;
    }
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float neg_step_size = (float)(-(double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const float w1 = 1.0f - beta1, w2 = 1.0f - beta2;
    for (long long i = threadIdx.x; i < count; i += kAdamThreads) {
        const float gi = grads[i] * grad_scale * clip;
        float m = exp_avg[i], v = exp_avg_sq[i];
        m = m + w1 * (gi - m);               // exp_avg.lerp_(grad, 1 - beta1)
        v = v * beta2 + w2 * gi * gi;        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        params[i] = params[i] + neg_step_size * m / denom;  // param.addcdiv_(exp_avg, denom, value=-step_size)
        exp_avg[i] = m;
        exp_avg_sq[i] = v;
    }
    if (loss_sums && losses && threadIdx.x < 6) losses[threadIdx.x] += loss_sums[threadIdx.x] * loss_scale;
}

static int check_update_args(const pfa_experience *ex, int64_t batch_rows, const pfa_mlp_dims *dims,
                             const pfa_ppo_hparams *hp) {
    PFA_REQUIRE(ex && hp, "ppo: null argument");
    PFA_REQUIRE(hp->num_minibatches >= 1 && hp->bptt_horizon >= 1, "ppo: bad minibatch partition");
    PFA_REQUIRE(batch_rows % hp->num_minibatches == 0, "batch_size must be divisible by minibatch_size");
    const int64_t mbs = batch_rows / hp->num_minibatches;
    PFA_REQUIRE(mbs % hp->bptt_horizon == 0, "minibatch_size must be divisible by bptt_horizon");
    (void)dims;
    return 0;
}

static int grad_grid(int64_t mb_rows) {
    const int64_t tiles = mb_rows / 16;
    const int64_t wgs = (tiles + kGradWaves - 1) / kGradWaves;
    return (int)(wgs < 256 ? (wgs < 1 ? 1 : wgs) : 256);
}

}  // namespace pfa

using namespace pfa;

extern "C" size_t pfa_ppo_workspace_bytes(const pfa_mlp_dims *dims, int64_t batch_rows, const pfa_ppo_hparams *hp) {
    if (!dims || !hp || hp->num_minibatches < 1) return 0;
    const int pp = mlp_offsets(dims->obs_stride, dims->num_actions).count + kNumStats;
    (void)batch_rows;
    const size_t partials = align_up((size_t)256 * pp * sizeof(float), 256);
    const size_t advp = align_up((size_t)hp->num_minibatches * kAdvChunks * 2 * sizeof(double), 256);
    return partials + advp;
}

extern "C" int pfa_ppo_adv_stats(const pfa_experience *exp, int64_t batch_rows, const pfa_ppo_hparams *hp, double *stats,
                                 void *workspace, pfa_stream_t stream) {
    if (int rc = check_update_args(exp, batch_rows, nullptr, hp)) return rc;
    PFA_REQUIRE(exp->advantages && stats && workspace, "ppo.adv_stats: null buffer");
    const int64_t mbs = batch_rows / hp->num_minibatches;
    // shares the workspace with pfa_ppo_mlp_grad: stream order keeps the two uses apart in time
    RowMap map{0, hp->num_minibatches, hp->bptt_horizon};
    double *partial = (double *)workspace;
    hipLaunchKernelGGL(adv_stats_partial_kernel, dim3(kAdvChunks, hp->num_minibatches), dim3(256), 0, (hipStream_t)stream,
                       exp->advantages, map, (long long)mbs, partial);
    PFA_LAUNCH_CHECK();
    hipLaunchKernelGGL(adv_stats_final_kernel, dim3((hp->num_minibatches + 63) / 64), dim3(64), 0, (hipStream_t)stream, partial,
                       hp->num_minibatches, stats);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_ppo_mlp_grad(const pfa_experience *exp, int64_t batch_rows, int32_t mb, const float *params,
                                const pfa_mlp_dims *dims, const pfa_ppo_hparams *hp, const double *adv_stats,
                                int64_t global_mb_rows, float *grads, void *workspace, pfa_stream_t stream) {
    if (int rc = check_update_args(exp, batch_rows, dims, hp)) return rc;
    PFA_REQUIRE(dims && dims->hidden == kHidden, "ppo.grad: hidden must be %d", kHidden);
    PFA_REQUIRE(dims->obs_stride == 16 || dims->obs_stride == 32 || dims->obs_stride == 64, "ppo.grad: obs_stride must be 16/32/64");
    PFA_REQUIRE(dims->num_actions >= 1 && dims->num_actions <= 15, "ppo.grad: num_actions must be in 1..15");
    PFA_REQUIRE(mb >= 0 && mb < hp->num_minibatches, "ppo.grad: minibatch index out of range");
    PFA_REQUIRE(exp->obs && exp->actions && exp->logprobs && exp->values && exp->advantages && exp->returns && params && grads &&
                    workspace,
                "ppo.grad: null buffer");
    PFA_REQUIRE(!hp->norm_adv || adv_stats, "ppo.grad: norm_adv needs adv_stats");
    const int64_t mbs = batch_rows / hp->num_minibatches;
    PFA_REQUIRE(mbs % 16 == 0, "ppo.grad: minibatch_size must be a multiple of 16 (got %lld)", (long long)mbs);
    PFA_REQUIRE(global_mb_rows >= mbs, "ppo.grad: global_mb_rows < local minibatch rows");
    const int pp = mlp_offsets(dims->obs_stride, dims->num_actions).count + kNumStats;
    const int grid = grad_grid(mbs);
    RowMap map{mb, hp->num_minibatches, hp->bptt_horizon};
    float *partials = (float *)workspace;
#define PFA_LAUNCH_GRAD(DPV)                                                                                               \
    {                                                                                                                      \
        constexpr size_t lds_bytes = (size_t)GradLds<DPV>::kFloats * sizeof(float);                                        \
        static_assert((size_t)(kHidden * DPV + kHidden + 16 * kHidden + 16 + kNumStats) * sizeof(float) <= lds_bytes,       \
                      "reduction buffer must fit in the tile area");                                                      \
        static bool attr_set = false;                                                                                      \
        if (!attr_set) {                                                                                                   \
            PFA_CHECK_HIP(hipFuncSetAttribute((const void *)ppo_mlp_grad_kernel<DPV>,                                      \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));                \
            attr_set = true;                                                                                               \
        }                                                                                                                  \
        hipLaunchKernelGGL(ppo_mlp_grad_kernel<DPV>, dim3(grid), dim3(kGradThreads), lds_bytes, (hipStream_t)stream, *exp,  \
                           map, (long long)mbs, params, dims->num_actions, *hp, adv_stats, (double)global_mb_rows, partials); \
    }
    {
        ScopedKernelTimer timer("ppo_mlp_grad", (hipStream_t)stream);
        switch (dims->obs_stride) {
            case 16: PFA_LAUNCH_GRAD(16) break;
            case 32: PFA_LAUNCH_GRAD(32) break;
            default: PFA_LAUNCH_GRAD(64) break;
        }
    }
#undef PFA_LAUNCH_GRAD
    PFA_LAUNCH_CHECK();
    ScopedKernelTimer timer2("ppo_reduce", (hipStream_t)stream);
    hipLaunchKernelGGL(ppo_reduce_kernel, dim3((pp + 31) / 32), dim3(256), 0, (hipStream_t)stream, partials, grid, pp, grads);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_adam_clip_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t count, float lr,
                                  float beta1, float beta2, float eps, int64_t step, float max_grad_norm, float grad_scale,
                                  const float *loss_sums, float *losses, float loss_scale, pfa_stream_t stream) {
    PFA_REQUIRE(params && grads && exp_avg && exp_avg_sq, "adam: null buffer");
    PFA_REQUIRE(count >= 1 && step >= 1, "adam: count and step must be >= 1");
    ScopedKernelTimer timer("adam_clip", (hipStream_t)stream);
    hipLaunchKernelGGL(adam_clip_kernel, dim3(1), dim3(kAdamThreads), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                       (long long)count, lr, beta1, beta2, eps, (long long)step, max_grad_norm, grad_scale, loss_sums, losses,
                       loss_scale);
    PFA_LAUNCH_CHECK();
    return 0;
}
