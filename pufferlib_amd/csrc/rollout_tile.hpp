// rollout_tile.hpp — the policy-forward + sampling tile code shared by every rollout kernel of the MLP policy
// (frameworks/cleanrl.py:60-66 Policy.forward with action=None = models.py:41-62 + cleanrl.py:25-47): the standalone
// forward (rollout.hip), the fused Squared rollout (rollout.hip) and the fused Stochastic rollout (stochastic.hip) run exactly
// this code, so the protocol path and every fused rollout give bit-identical numbers.
#pragma once
#include "common.hpp"
#include "lane_ops.hpp"
#include "mlp_tile.hpp"
#include "philox.hpp"
#include "sampler.hpp"

namespace pfa {

// ---------------------------------------------------------------------------------------------
// Shared by the standalone forward and the fused rollout: a 256-thread workgroup (4 wavefronts) owns one
// 16-row tile.  Wave w computes hidden tiles m in {2w, 2w+1} and its K-slice of the heads; after a barrier
// thread (le, lo) = (tid/16, tid%16) owns output `lo` of row `le`.  Both kernels run exactly this code, so
// policy(obs) through the protocol and the fused rollout give bit-identical numbers.
// ---------------------------------------------------------------------------------------------
constexpr int kRollThreads = 256;
constexpr int kRollWaves = 4;
constexpr int kMW = kMT / kRollWaves;  // hidden tiles per wave

// Where a Default policy's tensors live (models.py:24-39), so that the same tile code serves the kernel-layout buffer of the 128-wide
// policy (W1 rows padded to the observation stride) and a module's own tensors in torch's shapes (any width; general.py).
struct MlpView {
    const float *w1;   // encoder.weight [hidden][ldw1], columns >= cols read as 0
    int ldw1, cols;
    const float *b1;   // encoder.bias [hidden]
    const float *w2;   // decoder.weight [a][hidden]
    const float *b2;   // decoder.bias [a]
    const float *wv;   // value_head.weight [hidden]
    const float *bv;   // value_head.bias [1]
    int a, hidden;
    __device__ __forceinline__ float w2v(int o, int u) const { return o < a ? w2[o * hidden + u] : (o == a ? wv[u] : 0.0f); }
    __device__ __forceinline__ float b2v(int o) const { return o < a ? b2[o] : (o == a ? bv[0] : 0.0f); }
};
inline MlpView mlp_view_of_flat(const float *params, int dp, int a) {   // the flat layout of include/pufferlib_amd.h (hidden 128)
    const MlpOffsets off = mlp_offsets(dp, a);
    return MlpView{params + off.w1, dp, dp, params + off.b1, params + off.w2, params + off.b2, params + off.wv, params + off.bv, a, kHidden};
}

// KS = k-steps (4 observation columns each) the forward issues.  Default: the whole padded row.  A launch whose true feature
// count fits fewer steps passes the smaller number: the dropped products are pad weight x pad observation = +0 added to the
// accumulator (like the gradient kernel's KKU, ppo_update.hip), and every rollout-mode kernel of a policy takes the same KS,
// so the protocol path and the fused path stay bit-identical.
// MW = hidden tiles (16 units each) per wave: hidden = 64 MW, i.e. 64 / 128 / 256 / 512 for MW = 1 / 2 / 4 / 8.  The W1 fragments
// of any of them fit the registers of a one-wave-per-SIMD workgroup (MW x KS <= 128 floats at a 64-float row).
template <int DP, int KS = DP / 4, int MW = kMW>
struct SliceFrags {
    float w1f[MW][KS], b1f[MW][4], w2f[MW][4], bo[4];
    __device__ __forceinline__ void load(const float *params, int a) {
        const MlpOffsets off = mlp_offsets(DP, a);
        const int wv = wave_id(), c = lane_id() & 15, g = lane_id() >> 4;
#pragma unroll
        for (int i = 0; i < MW; ++i) {
            const int m = MW * wv + i;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) w1f[i][kk] = params[off.w1 + (16 * m + c) * DP + 4 * kk + g];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                b1f[i][r] = params[off.b1 + 16 * m + 4 * g + r];
                w2f[i][r] = w2v_at(params, off, a, c, 16 * m + 4 * g + r);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) bo[r] = wv == 0 ? b2v_at(params, off, a, 4 * g + r) : 0.0f;
    }
    __device__ __forceinline__ void load(const MlpView &p) {   // the same fragments from wherever the tensors live
        const int wv = wave_id(), c = lane_id() & 15, g = lane_id() >> 4;
#pragma unroll
        for (int i = 0; i < MW; ++i) {
            const int m = MW * wv + i;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) w1f[i][kk] = 4 * kk + g < p.cols ? p.w1[(size_t)(16 * m + c) * p.ldw1 + 4 * kk + g] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                b1f[i][r] = p.b1[16 * m + 4 * g + r];
                w2f[i][r] = p.w2v(c, 16 * m + 4 * g + r);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) bo[r] = wv == 0 ? p.b2v(4 * g + r) : 0.0f;
    }
};

// Global rows -> padded LDS tile, all 256 threads.
template <int DP>
__device__ __forceinline__ void stage_rows(const float *src, long long first_row, long long rows, float *xs,
                                           int tile_rows = 16) {
    constexpr int XS = XTile<DP>::XS, V = DP / 4;
    for (int idx = threadIdx.x; idx < 16 * V; idx += kRollThreads) {
        const int r = idx / V, c4 = idx - r * V;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < tile_rows && first_row + r < rows) x = *reinterpret_cast<const float4 *>(src + (first_row + r) * DP + 4 * c4);
        float2 *d = reinterpret_cast<float2 *>(xs + r * XS + 4 * c4);
        d[0] = make_float2(x.x, x.y);
        d[1] = make_float2(x.z, x.w);
    }
}

// LDS tile -> global rows (dst row r at dst + (first_row + r) * row_stride floats), all 256 threads.
template <int DP>
__device__ __forceinline__ void unstage_rows(const float *xs, float *dst, long long first_row, long long rows, size_t row_stride,
                                             int tile_rows = 16) {
    constexpr int XS = XTile<DP>::XS, V = DP / 4;
    for (int idx = threadIdx.x; idx < tile_rows * V; idx += kRollThreads) {
        const int r = idx / V, c4 = idx - r * V;
        if (first_row + r < rows) {
            const float2 *sp = reinterpret_cast<const float2 *>(xs + r * XS + 4 * c4);
            const float2 lo2 = sp[0], hi2 = sp[1];
            *reinterpret_cast<float4 *>(dst + (size_t)(first_row + r) * row_stride + 4 * c4) = make_float4(lo2.x, lo2.y, hi2.x, hi2.y);
        }
    }
}

// This wave's slice of models.Default.forward (models.py:41-62) for the 16 rows in xs -> part[wave][o * kPartStride + row].
template <int DP, int KS = DP / 4, int MW = kMW>   // MW = 1 or even
__device__ __forceinline__ void forward_slice(const SliceFrags<DP, KS, MW> &w, const float *xs, float (*part)[kPartFloats]) {
    constexpr int XS = XTile<DP>::XS;
    const int wv = wave_id(), c = lane_id() & 15, g = lane_id() >> 4;
    f32x4 h[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) h[i] = f32x4{w.b1f[i][0], w.b1f[i][1], w.b1f[i][2], w.b1f[i][3]};
    // every B value of the row first, the products behind them: left to itself the compiler sinks each ds_read next to its products
    // and drains lgkmcnt in front of every group (ISA of round 4: seven exposed LDS latencies per step of the rollout's dependent
    // chain); same products in the same order, so the numbers do not change
    float bv[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) bv[kk] = xs[c * XS + 4 * kk + g];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
        for (int i = 0; i < MW; ++i) h[i] = mfma16(w.w1f[i][kk], bv[kk], h[i]);
    }
    // the heads' contraction over this wave's 16 MW hidden units: two accumulator chains (even / odd tiles)
    f32x4 o0 = f32x4{w.bo[0], w.bo[1], w.bo[2], w.bo[3]}, o1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MW; i += 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o0 = mfma16(w.w2f[i][r], fmaxf(h[i][r], 0.0f), o0);
            if constexpr (MW >= 2) o1 = mfma16(w.w2f[i + 1][r], fmaxf(h[i + 1][r], 0.0f), o1);
        }
    }
    const f32x4 po = o0 + o1;  // partial out^T[o = 4g + r][row = c]
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wv][(4 * g + r) * kPartStride + c] = po[r];
}

// Sum of the four waves' head partials for (row le, output lo), then sample_row16 (sampler.hpp).
__device__ __forceinline__ LaneSample sample_lanes(const float (*part)[kPartFloats], int le, int lo, int a, float q) {
    const int i = lo * kPartStride + le;
    const float mine = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
    return sample_row16(mine, lo, a, q);
}
__device__ __forceinline__ LaneSample sample_lanes_heads(const float (*part)[kPartFloats], int le, int lo, int a, uint32_t heads, float q) {
    const int i = lo * kPartStride + le;
    const float mine = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
    return sample_row16_heads(mine, lo, a, heads, q);
}

}  // namespace pfa
