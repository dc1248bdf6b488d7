// sampler.hpp — sample_logits for one Discrete head (pufferlib/frameworks/cleanrl.py:25-47, action=None branch) on the
// 16 lanes of a DPP row: lane `lo` holds output `lo` of the row (`mine` = logit for lo < a, value for lo == a).
//   action = argmax(softmax(logits) / q)  == torch.multinomial(softmax(logits), 1) given its Exp(1) draw q.
// Shared by the MLP and LSTM policies so every path samples with identical arithmetic.
#pragma once
#include "common.hpp"
#include "lane_ops.hpp"
#include "philox.hpp"

namespace pfa {

struct LaneSample {
    int action;
    float logprob, entropy, value;
};

__device__ __forceinline__ LaneSample sample_row16(float mine, int lo, int a, float q) {
    const bool is_logit = lo < a;
    const float mx = row16_max(is_logit ? mine : -INFINITY);
    const float ex_ = is_logit ? expf(mine - mx) : 0.0f;
    const float se = row16_sum(ex_);
    const float lse = mx + logf(se);
    // argmax of p/q with torch's first-index tie rule
    float best = is_logit ? (ex_ / se) / q : -INFINITY;
    int besti = lo;
    row16_argmax(best, besti);
    LaneSample r;
    r.action = besti;
    const float nl = mine - lse;
    r.logprob = row16_sum(lo == besti ? nl : 0.0f);
    r.entropy = row16_sum(is_logit ? -nl * expf(nl) : 0.0f);
    r.value = row16_sum(lo == a ? mine : 0.0f);
    return r;
}

// MultiDiscrete (cleanrl.py:31-44): `heads` packs the head sizes (pfa_mlp_dims.heads); one draw per head over its own
// logit columns, the choices packed like the sizes, log-probabilities and entropies summed over heads.
__host__ __device__ inline int heads_count(uint32_t heads, int a) {  // number of heads, or -1 if the sizes do not sum to a
    int n = 0, total = 0;
    for (; n < 8 && ((heads >> (4 * n)) & 15u) != 0; ++n) total += (int)((heads >> (4 * n)) & 15u);
    return (n < 8 ? (heads >> (4 * n)) == 0 : true) && total == a ? n : -1;
}

__device__ __forceinline__ LaneSample sample_row16_heads(float mine, int lo, int a, uint32_t heads, float q) {
    LaneSample r;
    r.action = 0;
    r.logprob = r.entropy = 0.0f;
    int start = 0;
    for (int h = 0; h < 8; ++h) {
        const int sz = (int)((heads >> (4 * h)) & 15u);
        if (sz == 0) break;  // uniform
        const bool member = lo >= start && lo < start + sz;
        const float mx = row16_max(member ? mine : -INFINITY);
        const float ex_ = member ? expf(mine - mx) : 0.0f;
        const float se = row16_sum(ex_);
        const float lse = mx + logf(se);
        float best = member ? (ex_ / se) / q : -INFINITY;
        int besti = lo;
        row16_argmax(best, besti);
        const float nl = mine - lse;
        r.action |= (besti - start) << (4 * h);
        r.logprob += row16_sum(lo == besti ? nl : 0.0f);
        r.entropy += row16_sum(member ? -nl * expf(nl) : 0.0f);
        start += sz;
    }
    r.value = row16_sum(lo == a ? mine : 0.0f);
    return r;
}

// Training-mode sample_logits (cleanrl.py:38-44, given actions) on the same 16 lanes: this lane's normalised logit and
// softmax probability, the entropy of the head it belongs to, whether it is its head's chosen column, and the row's summed
// log-probability / entropy.  heads == 0 is one Discrete(a) head.
struct Row16Eval {
    float nl, p, head_entropy, logprob, entropy;
    bool is_logit, chosen;
};
__device__ __forceinline__ Row16Eval eval_row16_heads(float mine, int lo, int a, uint32_t heads, int packed_action) {
    Row16Eval r;
    r.nl = r.p = r.head_entropy = r.logprob = r.entropy = 0.0f;
    r.is_logit = lo < a;
    r.chosen = false;
    if (heads == 0) heads = (uint32_t)a;  // a <= 15: one nibble
    int start = 0;
    for (int h = 0; h < 8; ++h) {
        const int sz = (int)((heads >> (4 * h)) & 15u);
        if (sz == 0) break;  // uniform
        const bool member = lo >= start && lo < start + sz;
        const float mx = row16_max(member ? mine : -INFINITY);
        const float ev = member ? expf(mine - mx) : 0.0f;
        const float se = row16_sum(ev);
        const float nl = mine - (mx + logf(se)), p = ev / se;
        const float he = row16_sum(member ? -nl * p : 0.0f);
        const int act = start + ((packed_action >> (4 * h)) & 15);
        r.logprob += row16_sum(lo == act ? nl : 0.0f);
        r.entropy += he;
        if (member) {
            r.nl = nl;
            r.p = p;
            r.head_entropy = he;
            r.chosen = lo == act;
        }
        start += sz;
    }
    return r;
}

// Exp(1) noise of (row, step, column lo): explicit tensor if given, else the Philox stream (philox.hpp).
__device__ __forceinline__ float noise_lane(const float *noise_row_ptr, uint64_t seed, uint64_t step, uint64_t row, int lo,
                                            int a) {
    if (lo >= a) return 1.0f;
    if (noise_row_ptr) return noise_row_ptr[lo];
    const u32x4 w = philox4x32_10((uint32_t)row, (uint32_t)(lo >> 2), (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed,
                                  (uint32_t)(seed >> 32));
    const uint32_t wsel = (lo & 3) == 0 ? w.x : (lo & 3) == 1 ? w.y : (lo & 3) == 2 ? w.z : w.w;
    return -logf(philox_uniform(wsel));
}

}  // namespace pfa
