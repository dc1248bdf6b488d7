// mlp_tile.hpp — fp32 MFMA building blocks for pufferlib.models.Default (models.py:24-62):
//   hidden = relu(obs @ W1^T + b1);  logits = hidden @ W2^T + b2;  value = hidden @ Wv^T + bv
// on 16-row tiles with v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains; the 1e-5 parity target rules out
// bf16/fp16 MFMA and gfx950 has no xf32).
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
// The rollout kernels' partial out^T tiles in LDS: [output o][row] with rows of 18 floats.  Thread (row le, output lo) reads
// part[w][lo * kPartStride + le]; with rows of 16 the 8 even (odd) outputs of a 32-lane group met on ONE bank (ds_read_b32 banks are
// word mod 32: 16 lo + le), an 8-way conflict on the dependent chain of every rollout step; 18 lo + le is conflict-free, and the
// fragment-order stores (4g + r) * 18 + c stay 2-way, which a ds_write_b32 absorbs (MI355X_MICROARCH.md, LDS).
constexpr int kPartStride = 18;
constexpr int kPartFloats = kOut * kPartStride;

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

// LDS row stride of a 16 x DP observation tile: +2 floats makes the B-fragment read
// xs[c*XS + 4kk + g] hit 32 distinct banks per 32-lane group (bank = 2c + g) and keeps rows 8-byte aligned.
template <int DP>
struct XTile {
    static constexpr int XS = DP + 2;
    static constexpr int kFloats = 16 * XS;
};

}  // namespace pfa
