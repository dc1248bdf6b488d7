// ppo_tile.hpp — device pieces of the fused PPO gradient kernel (ppo_update.hip), also used by the archived kernel-structure
// experiments under tools/experiments/:
//   RowMap         minibatch row -> flat env-major experience row (clean_pufferl.py:455-457)
//   ppo_loss_tile  the PPO loss of clean_pufferl.py:202-238 and d(loss)/d(out^T fragment) for the 16 rows of a tile
#pragma once
#include <cmath>

#ifndef PFA_LOSS_EXP_REPAIR_ALL
#define PFA_LOSS_EXP_REPAIR_ALL 0   // 1 = the argument-repaired exponential for the softmax numerators too (A/B timing)
#endif

#include "common.hpp"
#include "mlp_tile.hpp"

namespace pfa {

constexpr int kNumStats = 8;                // 6 used: pg, v, entropy, old_kl, kl, clipfrac

struct RowMap {  // minibatch row q -> flat env-major experience row (clean_pufferl.py:455-457)
    int mb, nmb, horizon;
    __device__ __forceinline__ long long flat(long long q) const {
        const long long k = q / horizon, h = q - k * horizon;
        return ((long long)mb + k * nmb) * horizon + h;
    }
    // 32-bit flat row of minibatch row q (batches stay far below 2^31 rows).
    __device__ __forceinline__ unsigned row32(unsigned q) const {
        const unsigned k = q / (unsigned)horizon, h = q - k * (unsigned)horizon;
        return ((unsigned)mb + k * (unsigned)nmb) * (unsigned)horizon + h;
    }
    // Flat row of row r (0..15) of 16-row tile `tile` (wave-uniform).  When bptt_horizon is a multiple of 16 a tile
    // never straddles a segment: `first` = row32(16*tile), computed ONCE per tile on the scalar unit, then + r.
    __device__ __forceinline__ unsigned tile_first(unsigned tile) const {
        return row32(__builtin_amdgcn_readfirstlane(tile) * 16u);
    }
    __device__ __forceinline__ unsigned tile_row(unsigned tile, unsigned first, int r, bool aligned) const {
        return aligned ? first + (unsigned)r : row32(tile * 16u + (unsigned)r);
    }
};

__device__ __forceinline__ void wave_lds_fence() {
    // LDS is in-order per wave; this only stops the compiler from moving LDS accesses across the hand-off
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Reductions over the four 16-lane groups g of a wavefront (the logits of a row are spread over lane groups in the out^T
// C fragment).  SWAP: gfx950's v_permlane16_swap / v_permlane32_swap (VALU, no LDS round trip): with both operands = x,
// permlane16_swap leaves {rows 0,0,2,2} and {rows 1,1,3,3}, permlane32_swap {lo,lo} and {hi,hi}; combining the pair gives
// every lane the same tree ((g0+g1)+(g2+g3)) the xor-16 / xor-32 shuffles produce, bit for bit.
template <bool SWAP>
__device__ __forceinline__ float gsum(float x) {
    if constexpr (SWAP) {
        auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        const float s = __uint_as_float(p[0]) + __uint_as_float(p[1]);
        auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
        return __uint_as_float(q[0]) + __uint_as_float(q[1]);
    } else {
        x += __shfl_xor(x, 16, 64);
        x += __shfl_xor(x, 32, 64);
        return x;
    }
}
template <bool SWAP>
__device__ __forceinline__ float gmax(float x) {
    if constexpr (SWAP) {
        auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        const float s = fmaxf(__uint_as_float(p[0]), __uint_as_float(p[1]));
        auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
        return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
    } else {
        x = fmaxf(x, __shfl_xor(x, 16, 64));
        x = fmaxf(x, __shfl_xor(x, 32, 64));
        return x;
    }
}

// exp / log of the loss on the hardware transcendentals (v_exp_f32 / v_log_f32, 1 ulp) with the argument's rounding repaired:
// five instructions instead of the library's ~15.  The loss is VALU work on the MFMA waves' issue port, so its
// instruction count is launch time.  exp(x) = 2^(t + e) with t = fl(x log2e) and e = the product's rounding error + x times the
// constant's low part; 2^(t+e) = 2^t (1 + e ln2) to first order (|e| < 2^-23 |t|).  Relative error ~2 ulp over the loss's range
// (x <= 0 in the softmax, |x| small in the ratio) — the library forms are not bit-identical to torch's vectorised ones either.
__device__ __forceinline__ float loss_exp(float x) {
    const float kL = 1.44269502162933349609375f, kLl = 1.92596298909109e-8f;   // log2(e) = kL + kLl
    const float t = x * kL;
    const float e = fmaf(x, kLl, fmaf(x, kL, -t));
    const float r = __builtin_amdgcn_exp2f(t);
    return fmaf(r, e * 0.693147182464599609375f, r);
}
// The softmax numerators exp(logit - max), x <= 0: without the repair — two instructions instead of six on the loss's dependent
// chain.  The argument's rounding then shows as a relative error of |x| * 4e-8 in the result, i.e. < 1e-6 on probabilities that are
// themselves < e^-20 and ~1e-8 on the ones that matter; measured on the bench shape: parameters after two updates move by 5e-8
// against the repaired form, the launch gets 0.8 us shorter (profiles/r03_grad_variants.txt).  The ratio keeps the repaired form.
__device__ __forceinline__ float loss_exp_softmax(float x) {
#if PFA_LOSS_EXP_REPAIR_ALL
    return loss_exp(x);
#else
    return __builtin_amdgcn_exp2f(x * 1.44269502162933349609375f);
#endif
}
__device__ __forceinline__ float loss_log(float x) {   // x in [1, 16] here (sum of exp(logit - max))
    const float l2 = __builtin_amdgcn_logf(x);           // v_log_f32 = log2
    return fmaf(l2, 0.693147182464599609375f, l2 * -1.904654323148236e-9f);   // ln2 = hi + lo
}

struct RowScalars {
    int action;
    float old_logprob, old_value, adv, ret, weight;  // weight 0 for rows of a padding tile
};

struct LossOut {
    f32x4 dout;
    float pg, v_loss, ent, neg_logratio, kl, clipped;
};

// PPO loss for the rows of one tile (clean_pufferl.py:202-238) and d(loss)/d(out^T fragment).
// MH: MultiDiscrete (cleanrl.py:31-44) — `heads` packs the head sizes, rs.action the per-head choices (pfa_mlp_dims.heads); the
// log-softmax, the chosen log-probability and the entropy are taken per head and summed.  The single-head instantiation is the
// code the headline workload runs, unchanged.
// PERM (one Discrete head of at most 11 actions): the outputs sit in permuted rows of the out^T fragment — logical output o in
// slot 4 (o / 3) + o % 3, i.e. lane group g holds outputs 3g .. 3g+2 in registers 0..2 and register 3 is padding everywhere — so
// that dh = dout . W2v contracts over three k-steps instead of four and every per-logit loop below runs three times, not four.
template <bool MH, bool SWAP = false, bool PERM = false>
__device__ __forceinline__ LossOut ppo_loss_tile(const f32x4 &out, const RowScalars &rs, int a, uint32_t heads, int g,
                                                 const pfa_ppo_hparams &hp, float adv_mean, float adv_rden, float inv_rows) {
    float nl[4], p[4], hent[4], ent = 0.0f, new_logprob = 0.0f, new_value = 0.0f;
    bool chosen[4];
    if constexpr (MH) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            nl[r] = p[r] = hent[r] = 0.0f;
            chosen[r] = false;
            if (4 * g + r == a) new_value = out[r];
        }
        int start = 0;
        for (int h = 0; h < 8; ++h) {
            const int sz = (int)((heads >> (4 * h)) & 15u);
            if (sz == 0) break;  // uniform
            bool mem[4];
            float lmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 4 * g + r;
                mem[r] = o >= start && o < start + sz;
                if (mem[r]) lmax = fmaxf(lmax, out[r]);
            }
            lmax = fmaxf(lmax, __shfl_xor(lmax, 16, 64));
            lmax = fmaxf(lmax, __shfl_xor(lmax, 32, 64));
            float ev[4], se = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ev[r] = mem[r] ? loss_exp(out[r] - lmax) : 0.0f;
                se += ev[r];
            }
            se += __shfl_xor(se, 16, 64);
            se += __shfl_xor(se, 32, 64);
            const float lse = lmax + loss_log(se), inv_se = __builtin_amdgcn_rcpf(se);
            const int act = start + (int)(((uint32_t)rs.action >> (4 * h)) & 15u);
            float he = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (mem[r]) {
                    nl[r] = out[r] - lse;
                    p[r] = ev[r] * inv_se;
                    he -= nl[r] * p[r];
                    chosen[r] = 4 * g + r == act;
                    if (chosen[r]) new_logprob += nl[r];
                }
            he += __shfl_xor(he, 16, 64);
            he += __shfl_xor(he, 32, 64);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (mem[r]) hent[r] = he;
            ent += he;
            start += sz;
        }
        new_logprob += __shfl_xor(new_logprob, 16, 64);
        new_logprob += __shfl_xor(new_logprob, 32, 64);
        new_value += __shfl_xor(new_value, 16, 64);
        new_value += __shfl_xor(new_value, 32, 64);
    } else {
    // log-softmax over the A logits of row c, spread over lane groups: reduce with xor 16 / 32
    static_assert(!(MH && PERM), "the permuted layout is the single-head form");
    constexpr int NR = PERM ? 3 : 4;
    const int o0 = PERM ? 3 * g : 4 * g;     // logical output of register 0
#pragma unroll
    for (int r = NR; r < 4; ++r) {
        nl[r] = p[r] = 0.0f;
        chosen[r] = false;
    }
    float lmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < NR; ++r)
        if (o0 + r < a) lmax = fmaxf(lmax, out[r]);
    lmax = gmax<SWAP>(lmax);
    float ev[4], se = 0.0f;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        ev[r] = o0 + r < a ? loss_exp_softmax(out[r] - lmax) : 0.0f;
        se += ev[r];
    }
    se = gsum<SWAP>(se);
    const float lse = lmax + loss_log(se);
    const float inv_se = __builtin_amdgcn_rcpf(se);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int o = o0 + r;
        nl[r] = out[r] - lse;
        p[r] = ev[r] * inv_se;  // softmax; the exponentials are shared with the log-sum-exp
        if (o < a) ent -= nl[r] * p[r];
        chosen[r] = o == rs.action;
        if (chosen[r]) new_logprob = nl[r];
        if (o == a) new_value = out[r];
    }
    ent = gsum<SWAP>(ent);
    new_logprob = gsum<SWAP>(new_logprob);
    new_value = gsum<SWAP>(new_value);
#pragma unroll
    for (int r = 0; r < 4; ++r) hent[r] = ent;
    }

    const float logratio = new_logprob - rs.old_logprob;
    const float ratio = loss_exp(logratio);
    const float adv = hp.norm_adv ? (rs.adv - adv_mean) * adv_rden : rs.adv;   // adv_rden = 1 / (std + 1e-8), taken once per launch
    const float lo = 1.0f - hp.clip_coef, hi = 1.0f + hp.clip_coef;
    const float pg1 = -adv * ratio;
    const float pg2 = -adv * fminf(fmaxf(ratio, lo), hi);
    const bool inside = ratio >= lo && ratio <= hi;
    // d pg / d ratio under torch.max's tie rule (grad/2 to each side) and clamp's pass-through inside [lo, hi]
    float dpg_dratio;
    if (pg1 > pg2) dpg_dratio = -adv;
    else if (pg1 < pg2) dpg_dratio = inside ? -adv : 0.0f;
    else dpg_dratio = inside ? -adv : -0.5f * adv;
    const float scale = inv_rows * rs.weight;
    const float g_lp = dpg_dratio * ratio * scale;  // d loss / d new_logprob

    float v_loss, dv;
    if (hp.clip_vloss) {
        const float du = new_value - rs.ret;
        const float vl_u = du * du;
        const float delta = new_value - rs.old_value;
        const float v_clipped = rs.old_value + fminf(fmaxf(delta, -hp.vf_clip_coef), hp.vf_clip_coef);
        const float dc = v_clipped - rs.ret;
        const float vl_c = dc * dc;
        const bool vin = delta >= -hp.vf_clip_coef && delta <= hp.vf_clip_coef;
        v_loss = 0.5f * fmaxf(vl_u, vl_c);
        const float gu = 2.0f * du, gc = vin ? 2.0f * dc : 0.0f;
        const float sel = vl_u > vl_c ? gu : (vl_u < vl_c ? gc : 0.5f * (gu + gc));
        dv = 0.5f * sel;
    } else {
        const float du = new_value - rs.ret;
        v_loss = 0.5f * du * du;
        dv = du;
    }
    dv *= hp.vf_coef * scale;

    LossOut lo_;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = PERM ? (r < 3 ? 3 * g + r : 99) : 4 * g + r;
        float d = 0.0f;
        if (o < a) {
            // d new_logprob/d logit_o = [o==action] - p_o ; d entropy/d logit_o = -p_o (nl_o + H)
            d = g_lp * ((chosen[r] ? 1.0f : 0.0f) - p[r]) + hp.ent_coef * scale * p[r] * (nl[r] + hent[r]);
        } else if (o == a) {
            d = dv;
        }
        lo_.dout[r] = d;
    }
    lo_.pg = fmaxf(pg1, pg2) * rs.weight;
    lo_.v_loss = v_loss * rs.weight;
    lo_.ent = ent * rs.weight;
    lo_.neg_logratio = -logratio * rs.weight;
    lo_.kl = ((ratio - 1.0f) - logratio) * rs.weight;
    lo_.clipped = (fabsf(ratio - 1.0f) > hp.clip_coef ? 1.0f : 0.0f) * rs.weight;
    return lo_;
}

// The loss sums at the end of the gradient bucket are (hi, lo) float pairs of f64 sums: an f32 all-reduce of the bucket keeps ~48 bits.
constexpr int kTailFloats = 2 * kNumStats;


}  // namespace pfa
