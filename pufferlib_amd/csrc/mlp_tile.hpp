// mlp_tile.hpp — fp32 MFMA building blocks for pufferlib.models.Default (models.py:24-62):
//   hidden = relu(obs @ W1^T + b1);  logits = hidden @ W2^T + b2;  value = hidden @ Wv^T + bv
// on 16-row tiles with v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains; the 1e-5 parity target rules out
// bf16/fp16 MFMA and gfx950 has no xf32).  One wavefront owns one 16-row tile end to end.
//
// Fragment conventions of __builtin_amdgcn_mfma_f32_16x16x4f32 (c = lane&15, g = lane>>4):
//   A[i][k]: lane supplies A[i=c][k=g]     B[k][j]: lane supplies B[k=g][j=c]
//   C[i][j]: lane holds  C[i=4g+r][j=c], r = 0..3
// Everything is computed TRANSPOSED so that one product's C fragment is the next product's B fragment
// with no data movement (a K-permutation is free as long as A uses the same one):
//   hidden^T[u][row] = W1[u][:] . X[row][:]      A = W1 frags (registers), B = X tile (LDS)
//   out^T[o][row]    = W2v[o][:] . hidden^T[:][row]   B = relu(hidden^T) C-fragments, k-slot g <-> u = 16m+4g+r
// where W2v stacks decoder rows (o < A), the value head (o == A) and zero rows up to 16.
#pragma once
#include "common.hpp"

namespace pfa {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kHidden = 128;
constexpr int kMT = kHidden / 16;  // hidden tiles
constexpr int kOut = 16;           // padded head rows

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Flat parameter layout (include/pufferlib_amd.h).
struct MlpOffsets {
    int w1, b1, w2, b2, wv, bv, count;
};
__host__ __device__ inline MlpOffsets mlp_offsets(int dp, int a) {
    MlpOffsets o;
    o.w1 = 0;
    o.b1 = kHidden * dp;
    o.w2 = o.b1 + kHidden;
    o.b2 = o.w2 + a * kHidden;
    o.wv = o.b2 + a;
    o.bv = o.wv + kHidden;
    o.count = o.bv + 1;
    return o;
}

__device__ __forceinline__ float w2v_at(const float *params, const MlpOffsets &off, int a, int o, int u) {
    if (o < a) return params[off.w2 + o * kHidden + u];
    if (o == a) return params[off.wv + u];
    return 0.0f;
}
__device__ __forceinline__ float b2v_at(const float *params, const MlpOffsets &off, int a, int o) {
    if (o < a) return params[off.b2 + o];
    if (o == a) return params[off.bv];
    return 0.0f;
}

// Per-lane register fragments of the forward weights.
template <int DP>
struct MlpFwdFrags {
    float w1[kMT][DP / 4];  // W1[16m + c][4kk + g]
    float b1[kMT][4];       // b1[16m + 4g + r]      (accumulator init)
    float w2[kMT][4];       // W2v[o = c][16m + 4g + r]
    float bo[4];            // b2v[o = 4g + r]

    __device__ __forceinline__ void load(const float *params, int a) {
        const MlpOffsets off = mlp_offsets(DP, a);
        const int c = lane_id() & 15, g = lane_id() >> 4;
#pragma unroll
        for (int m = 0; m < kMT; ++m) {
#pragma unroll
            for (int kk = 0; kk < DP / 4; ++kk) w1[m][kk] = params[off.w1 + (16 * m + c) * DP + 4 * kk + g];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                b1[m][r] = params[off.b1 + 16 * m + 4 * g + r];
                w2[m][r] = w2v_at(params, off, a, c, 16 * m + 4 * g + r);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) bo[r] = b2v_at(params, off, a, 4 * g + r);
    }
};

// LDS row stride of a 16 x DP observation tile: +2 floats makes the B-fragment read
// xs[c*XS + 4kk + g] hit 32 distinct banks per 32-lane group (bank = 2c + g) and keeps rows 8-byte aligned.
template <int DP>
struct XTile {
    static constexpr int XS = DP + 2;
    static constexpr int kFloats = 16 * XS;
};

// hidden^T (post-ReLU) in h[m] and out^T in `out` for the 16 rows staged in xs.
template <int DP>
__device__ __forceinline__ void mlp_forward_tile(const MlpFwdFrags<DP> &w, const float *xs, f32x4 (&h)[kMT], f32x4 &out) {
    constexpr int XS = XTile<DP>::XS;
    const int c = lane_id() & 15, g = lane_id() >> 4;
#pragma unroll
    for (int m = 0; m < kMT; ++m) h[m] = f32x4{w.b1[m][0], w.b1[m][1], w.b1[m][2], w.b1[m][3]};
#pragma unroll
    for (int kk = 0; kk < DP / 4; ++kk) {
        const float b = xs[c * XS + 4 * kk + g];
#pragma unroll
        for (int m = 0; m < kMT; ++m) h[m] = mfma16(w.w1[m][kk], b, h[m]);
    }
    f32x4 o0 = f32x4{w.bo[0], w.bo[1], w.bo[2], w.bo[3]};
    f32x4 o1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < kMT; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) h[m][r] = fmaxf(h[m][r], 0.0f);
    }
    // two accumulators so consecutive head MFMAs are independent (40-cycle dependent latency vs 32 issue)
#pragma unroll
    for (int m = 0; m < kMT; m += 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o0 = mfma16(w.w2[m][r], h[m][r], o0);
            o1 = mfma16(w.w2[m + 1][r], h[m + 1][r], o1);
        }
    }
    out = o0 + o1;
}

// Gather the A logits and the value of row `c` (spread over lane groups in out^T) into lanes 0..15.
// Valid for A <= 15.  Lanes >= 16 receive garbage.
__device__ __forceinline__ void gather_row_outputs(const f32x4 &out, int a, float (&logits)[15], float &value) {
    const int lane = lane_id();
    float vals[16];
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
#pragma unroll
        for (int r = 0; r < 4; ++r) vals[4 * gg + r] = __shfl(out[r], (lane & 15) + 16 * gg, 64);
    }
    value = 0.0f;
#pragma unroll
    for (int o = 0; o < 15; ++o) logits[o] = vals[o];
#pragma unroll
    for (int o = 0; o < 16; ++o)
        if (o == a) value = vals[o];
}

// sample_logits for one Discrete head (frameworks/cleanrl.py:25-47), action=None branch:
//   action = argmax(softmax(logits) / q)  == torch.multinomial(softmax(logits), 1) given its Exp(1) draw q.
struct SampleOut {
    int action;
    float logprob, entropy;
};
__device__ __forceinline__ SampleOut sample_logits_row(const float (&logits)[15], int a, const float (&q)[15]) {
    float mx = -INFINITY;
#pragma unroll
    for (int o = 0; o < 15; ++o)
        if (o < a) mx = fmaxf(mx, logits[o]);
    float e[15], se = 0.0f;
#pragma unroll
    for (int o = 0; o < 15; ++o) {
        e[o] = o < a ? expf(logits[o] - mx) : 0.0f;
        se += e[o];
    }
    const float lse = mx + logf(se);
    SampleOut s;
    s.action = 0;
    float best = -INFINITY, ent = 0.0f;
    s.logprob = 0.0f;
#pragma unroll
    for (int o = 0; o < 15; ++o) {
        if (o < a) {
            const float p = e[o] / se;
            const float score = p / q[o];
            if (score > best) {
                best = score;
                s.action = o;
            }
            const float nl = logits[o] - lse;
            ent -= nl * expf(nl);
        }
    }
    float lp = 0.0f;
#pragma unroll
    for (int o = 0; o < 15; ++o)
        if (o == s.action) lp = logits[o] - lse;
    s.logprob = lp;
    s.entropy = ent;
    return s;
}

}  // namespace pfa
