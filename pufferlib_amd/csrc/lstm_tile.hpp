// lstm_tile.hpp — fp32 MFMA building blocks of the recurrent policy (pufferlib.models.LSTMWrapper, models.py:64-111):
//   xe = relu(obs @ W1^T + b1)                                   Default.encode_observations (models.py:41-50)
//   gates = [xe | h] @ [W_ih | W_hh]^T + (b_ih + b_hh)           nn.LSTM(128, 128), gate order i, f, g, o
//   c' = sig(f) c + sig(i) tanh(g);  h' = sig(o) tanh(c')
//   logits, value = h' @ W2v^T + b2v                             Default.decode_actions (models.py:52-62)
// for 16-row tiles owned by a 4-wave workgroup.  Like mlp_tile.hpp everything is computed transposed
// (out^T[col][row] = W[col][:] . in[row][:]) so the weights are the MFMA A operand and a row tile in LDS is the B operand.
//
// The gate matrix Wcat = [W_ih | W_hh] (512 x 256 fp32 = 512 KB) does not fit in registers or LDS, so every step
// streams it from L2 as pre-packed A fragments (lstm_pack_kernel): wave w owns, for every gate q, the hidden units
// 32w .. 32w+31, i.e. the 8 column tiles ct = 2q + sub, col(w, ct, i) = 128q + 32w + 16sub + i.  After the product
// lane (c, g) holds for batch row c the four gates of units 32w + 16sub + 4g + r — the cell update is register-local.
// The contraction index is permuted, k(kq, s, g) = 16kq + 4g + s for k-step 4kq + s, so that one ds_read_b128 of the
// [row][k] tile and one 16-byte global load of the packed weights feed four consecutive MFMAs.
#pragma once
#include "common.hpp"
#include "mlp_tile.hpp"

namespace pfa {

constexpr int kLH = 128;          // hidden units
constexpr int kLG = 4 * kLH;      // gate columns
constexpr int kLK = 2 * kLH;      // contraction length of the gate product: [xe | h]
constexpr int kXHS = kLK + 4;     // LDS row stride of an [row][xe | h] tile (floats); rows stay 16-byte aligned
constexpr int kXHTile = 16 * kXHS;
constexpr int kGatePackFloats = kLG * kLK;
constexpr int kLstmThreads = 256;  // 4 wavefronts per tile workgroup

struct LstmOffsets {
    int w_ih, w_hh, b_ih, b_hh, count;
};
__host__ __device__ inline LstmOffsets lstm_offsets(int dp, int a) {
    LstmOffsets o;
    o.w_ih = mlp_offsets(dp, a).count;
    o.w_hh = o.w_ih + kLG * kLH;
    o.b_ih = o.w_hh + kLG * kLH;
    o.b_hh = o.b_ih + kLG;
    o.count = o.b_hh + kLG;
    return o;
}

__device__ __forceinline__ int gate_col(int w, int ct, int i) { return 128 * (ct >> 1) + 32 * w + 16 * (ct & 1) + i; }

// Gate activations on the hardware transcendentals (v_exp_f32 / v_rcp_f32, 1 ulp each): four instructions instead of the ~25 of
// expf + IEEE division — the cell update is VALU work that does not overlap the MFMAs, 3.5 us per step and wave with
// the library forms.  sigmoid(x) = 1 / (1 + 2^(-x log2 e)); saturates cleanly (2^(+big) = inf -> 0, 2^(-big) = 0 -> 1).
// tanh(x) = 2 sigmoid(2x) - 1 (absolute error ~1e-7; it only ever multiplies a gate).  Well inside the 1e-5 parity bound: the
// reference's own vectorised sigmoid / tanh differ from libm by as much.
__device__ __forceinline__ float sigmoid_f(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}
__device__ __forceinline__ float tanh_f(float x) { return fmaf(2.0f, sigmoid_f(2.0f * x), -1.0f); }

// Register-resident weights of one wave (everything except the gate matrix).
template <int DP>
struct LstmFrags {
    float w1f[2][DP / 4];  // encoder tiles m = 2w + i: W1[16m + c][4kk + g]
    float b1f[2][4];       // b1[16m + 4g + r]
    float w2f[2][4];       // heads, this wave's K slice: W2v[o = c][32w + 16j + 4g + s]
    float bo[4];           // head bias of outputs 4g + r (wave 0 only, the partials are summed)
    __device__ __forceinline__ void load(const float *params, int a) {
        const MlpOffsets off = mlp_offsets(DP, a);
        const int wv = wave_id(), c = lane_id() & 15, g = lane_id() >> 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = 2 * wv + i;
#pragma unroll
            for (int kk = 0; kk < DP / 4; ++kk) w1f[i][kk] = params[off.w1 + (16 * m + c) * DP + 4 * kk + g];
#pragma unroll
            for (int r = 0; r < 4; ++r) b1f[i][r] = params[off.b1 + 16 * m + 4 * g + r];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s) w2f[j][s] = w2v_at(params, off, a, c, 32 * wv + 16 * j + 4 * g + s);
#pragma unroll
        for (int r = 0; r < 4; ++r) bo[r] = wv == 0 ? b2v_at(params, off, a, 4 * g + r) : 0.0f;
    }
};

// One float4 of the forward gate pack: dst index ((w*16 + kq)*8 + ct)*64 + lane  <-  Wcat[col(w, ct, c)][16kq + 4g .. +3]
__device__ __forceinline__ float4 lstm_pack_fwd_elem(const float *__restrict__ params, int dp, int a, int idx) {
    const LstmOffsets lo = lstm_offsets(dp, a);
    const int lane = idx & 63, ct = (idx >> 6) & 7, kq = (idx >> 9) & 15, w = idx >> 13;
    const int c = lane & 15, g = lane >> 4;
    const int col = gate_col(w, ct, c), k0 = 16 * kq + 4 * g;
    const float *src = k0 < kLH ? params + lo.w_ih + col * kLH + k0 : params + lo.w_hh + col * kLH + (k0 - kLH);
    return make_float4(src[0], src[1], src[2], src[3]);  // the flat parameter vector is only 4-byte aligned here
}

// b_ih + b_hh -> LDS [512], all threads of the workgroup (the cell update reads it as float4 per column tile).
__device__ __forceinline__ void stage_gate_bias(const float *params, int dp, int a, float *gbias) {
    const LstmOffsets lo = lstm_offsets(dp, a);
    for (int i = threadIdx.x; i < kLG; i += blockDim.x) gbias[i] = params[lo.b_ih + i] + params[lo.b_hh + i];
}

// encode_observations for the 16 rows in xs ([16][DP+2]) -> x half of the xh tile; wave w writes units 32w..32w+31.
template <int DP>
__device__ __forceinline__ void lstm_encode(const LstmFrags<DP> &w, const float *xs, float *xh, float *xe_row = nullptr) {
    constexpr int XS = XTile<DP>::XS;
    const int wv = wave_id(), c = lane_id() & 15, g = lane_id() >> 4;
    f32x4 h0 = f32x4{w.b1f[0][0], w.b1f[0][1], w.b1f[0][2], w.b1f[0][3]};
    f32x4 h1 = f32x4{w.b1f[1][0], w.b1f[1][1], w.b1f[1][2], w.b1f[1][3]};
    // the B values of the row run eight k-steps ahead of their products (two register chunks): the compiler would otherwise sink
    // every ds_read next to its two products and expose the LDS latency DP / 4 times; same products, same order
    constexpr int KS = DP / 4, CH = 8, NCH = (KS + CH - 1) / CH;
    float bv[2][CH];
#pragma unroll
    for (int q = 0; q < CH; ++q) bv[0][q] = q < KS ? xs[c * XS + 4 * q + g] : 0.0f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if (ch + 1 < NCH) {
#pragma unroll
            for (int q = 0; q < CH; ++q) bv[(ch + 1) & 1][q] = (ch + 1) * CH + q < KS ? xs[c * XS + 4 * ((ch + 1) * CH + q) + g] : 0.0f;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int kk = ch * CH + q;
            if (kk < KS) {
                h0 = mfma16(w.w1f[0][kk], bv[ch & 1][q], h0);
                h1 = mfma16(w.w1f[1][kk], bv[ch & 1][q], h1);
            }
        }
    }
    float4 *dst = reinterpret_cast<float4 *>(xh + c * kXHS + 32 * wv + 4 * g);
    dst[0] = make_float4(fmaxf(h0[0], 0.f), fmaxf(h0[1], 0.f), fmaxf(h0[2], 0.f), fmaxf(h0[3], 0.f));
    dst[4] = make_float4(fmaxf(h1[0], 0.f), fmaxf(h1[1], 0.f), fmaxf(h1[2], 0.f), fmaxf(h1[3], 0.f));
    if (xe_row) {  // this lane's row of the [rows][128] encoder output kept for the backward pass
        float4 *gx = reinterpret_cast<float4 *>(xe_row + 32 * wv + 4 * g);
        gx[0] = dst[0];
        gx[4] = dst[4];
    }
}

// ---------------------------------------------------------------------------------------------
// Streaming product: acc[rt][a] += sum over G groups and 4 k-steps of  A(gq, a, s) * B(rt, gq, s)
//   A: packed fragments streamed from L2, float4 (= 4 k-steps) per lane at wp[(gq*NA + a)*64 + lane];
//   B: LDS tile rows, float4 at brow + rt*BT + 16*gq  (brow already includes the lane's row and 4g offset).
// The weight stream runs D-1 groups ahead through a D-deep register ring (ring_prefetch issues the first D-1 groups; call
// it early so their L2 latency hides behind the preceding phase).  `side(gq)` runs once per group right after that group's
// weight loads are issued: callers spread their own global loads / stores over the product with it, which is how memory
// phases overlap the MFMAs inside one instruction stream (gfx9 retires loads and stores through ONE in-order vmcnt, so a
// burst of stores in front of the ring would stall it; a few per group, D-1 groups of slack, does not).
// The loop is fully unrolled with a scheduling barrier per group so the compiler neither hoists the stream wholesale
// (register blow-up) nor sinks the side operations.
// ---------------------------------------------------------------------------------------------
// One fragment of the packed stream: wave-uniform base (SGPR pair) + compile-time byte offset + 32-bit lane offset, so
// the load uses the scalar-base addressing form and no per-load 64-bit address registers.
struct Stream {
    const char *base;  // wave-uniform
    unsigned loff;     // lane * 16
};
// Call once per step: the lane offset is laundered through an empty asm so the 128 per-fragment addresses are formed
// where they are used (scalar base + immediate) instead of being hoisted out of the step loop into 256 registers.
__device__ __forceinline__ Stream stream_begin(const float4 *__restrict__ wp) {
    unsigned loff = (unsigned)lane_id() * 16u;
    asm volatile("" : "+v"(loff));
    return {reinterpret_cast<const char *>(wp), loff};
}
__device__ __forceinline__ float4 stream_ld(const Stream &st, int frag) {
    return *reinterpret_cast<const float4 *>(st.base + (size_t)frag * 1024 + st.loff);
}

template <int NA, int D>
__device__ __forceinline__ void ring_prefetch(const Stream &wp, float4 (&ring)[D][NA]) {
#pragma unroll
    for (int j = 0; j < D - 1; ++j)
#pragma unroll
        for (int a = 0; a < NA; ++a) ring[j][a] = stream_ld(wp, j * NA + a);
}

__device__ __forceinline__ float f4_at(const float4 &v, int s) { return s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w; }

template <int G, int NA, int RT, int BT, int D, class Side>
__device__ __forceinline__ void stream_product(const Stream &wp, const float *brow, float4 (&ring)[D][NA],
                                               f32x4 (&acc)[RT][NA], Side &&side) {
    static_assert((D & (D - 1)) == 0, "ring depth must be a power of two");
    float4 bc[RT], bn[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) bc[rt] = *reinterpret_cast<const float4 *>(brow + rt * BT);
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
        if (gq + D - 1 < G) {
#pragma unroll
            for (int a = 0; a < NA; ++a) ring[(gq + D - 1) & (D - 1)][a] = stream_ld(wp, (gq + D - 1) * NA + a);
        }
        side(gq);
        if (gq + 1 < G) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) bn[rt] = *reinterpret_cast<const float4 *>(brow + rt * BT + 16 * (gq + 1));
        }
        // s outermost: consecutive MFMAs hit different accumulators (no back-to-back dependent issue)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rt][a] = mfma16(f4_at(ring[gq & (D - 1)][a], s4), f4_at(bc[rt], s4), acc[rt][a]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) bc[rt] = bn[rt];
        __builtin_amdgcn_sched_barrier(0);
    }
}

struct NoSide {
    __device__ __forceinline__ void operator()(int) const {}
};

constexpr int kGateRing = 4;  // ring depth of the forward gate product (8 fragments per group)

// The cell update on one row tile's gate products (+ the bias from LDS, gbias[512]); cst = c state of units (sub, r).
// Returns h' and leaves the activations (i, f, g, o) in acc (what the backward pass stores).
__device__ __forceinline__ void lstm_cell(f32x4 (&acc)[8], const float *gbias, f32x4 (&cst)[2], f32x4 (&hout)[2]) {
    const int wv = wave_id(), g = lane_id() >> 4;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
        const float4 b = *reinterpret_cast<const float4 *>(gbias + gate_col(wv, ct, 4 * g));
        acc[ct] += f32x4{b.x, b.y, b.z, b.w};
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float i = sigmoid_f(acc[0 + sub][r]);
            const float f = sigmoid_f(acc[2 + sub][r]);
            const float gg = tanh_f(acc[4 + sub][r]);
            const float o = sigmoid_f(acc[6 + sub][r]);
            const float cn = f * cst[sub][r] + i * gg;
            acc[0 + sub][r] = i;
            acc[2 + sub][r] = f;
            acc[4 + sub][r] = gg;
            acc[6 + sub][r] = o;
            cst[sub][r] = cn;
            hout[sub][r] = o * tanh_f(cn);
        }
}

// This wave's K slice of decode_actions on the h half of an xh tile -> part[wave][o * kPartStride + row].
template <int DP>
__device__ __forceinline__ void lstm_heads(const LstmFrags<DP> &w, const float *xh, float (*part)[kPartFloats]) {
    const int wv = wave_id(), c = lane_id() & 15, g = lane_id() >> 4;
    f32x4 o0 = f32x4{w.bo[0], w.bo[1], w.bo[2], w.bo[3]};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float4 b = *reinterpret_cast<const float4 *>(xh + c * kXHS + kLH + 32 * wv + 16 * j + 4 * g);
        o0 = mfma16(w.w2f[j][0], b.x, o0);
        o0 = mfma16(w.w2f[j][1], b.y, o0);
        o0 = mfma16(w.w2f[j][2], b.z, o0);
        o0 = mfma16(w.w2f[j][3], b.w, o0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wv][(4 * g + r) * kPartStride + c] = o0[r];
}

// c state <-> global [rows][128] in the cell layout (lane (c, g): row c, units 32w + 16sub + 4g + r).
__device__ __forceinline__ void load_cstate(const float *cglob, long long row, bool ok, f32x4 (&cst)[2]) {
    const int wv = wave_id(), g = lane_id() >> 4;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = *reinterpret_cast<const float4 *>(cglob + row * kLH + 32 * wv + 16 * sub + 4 * g);
        cst[sub] = f32x4{v.x, v.y, v.z, v.w};
    }
}
__device__ __forceinline__ void store_cstate(float *cglob, long long row, bool ok, const f32x4 (&cst)[2]) {
    const int wv = wave_id(), g = lane_id() >> 4;
    if (!ok) return;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
        *reinterpret_cast<float4 *>(cglob + row * kLH + 32 * wv + 16 * sub + 4 * g) =
            make_float4(cst[sub][0], cst[sub][1], cst[sub][2], cst[sub][3]);
}
// h state: global [rows][128] <-> the h half of an xh tile, all 256 threads.
__device__ __forceinline__ void load_hstate(const float *hglob, long long first_row, long long rows, float *xh) {
    for (int idx = threadIdx.x; idx < 16 * (kLH / 4); idx += kLstmThreads) {
        const int r = idx / (kLH / 4), c4 = idx % (kLH / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (hglob && first_row + r < rows) v = *reinterpret_cast<const float4 *>(hglob + (first_row + r) * kLH + 4 * c4);   // (null: zero state)
        *reinterpret_cast<float4 *>(xh + r * kXHS + kLH + 4 * c4) = v;
    }
}
__device__ __forceinline__ void store_hstate(float *hglob, long long first_row, long long rows, const float *xh) {
    for (int idx = threadIdx.x; idx < 16 * (kLH / 4); idx += kLstmThreads) {
        const int r = idx / (kLH / 4), c4 = idx % (kLH / 4);
        if (first_row + r < rows)
            *reinterpret_cast<float4 *>(hglob + (first_row + r) * kLH + 4 * c4) =
                *reinterpret_cast<const float4 *>(xh + r * kXHS + kLH + 4 * c4);
    }
}

// Same staging as rollout.hip's stage_rows / unstage_rows (float2 pieces: the xs stride is DP+2).
template <int DP>
__device__ __forceinline__ void lstm_stage_obs(const float *src, long long first_row, long long rows, float *xs) {
    constexpr int XS = XTile<DP>::XS, V = DP / 4;
    for (int idx = threadIdx.x; idx < 16 * V; idx += kLstmThreads) {
        const int r = idx / V, c4 = idx - r * V;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (first_row + r < rows) x = *reinterpret_cast<const float4 *>(src + (first_row + r) * DP + 4 * c4);
        float2 *d = reinterpret_cast<float2 *>(xs + r * XS + 4 * c4);
        d[0] = make_float2(x.x, x.y);
        d[1] = make_float2(x.z, x.w);
    }
}
template <int DP>
__device__ __forceinline__ void lstm_unstage_obs(const float *xs, float *dst, long long first_row, long long rows,
                                                 size_t row_stride) {
    constexpr int XS = XTile<DP>::XS, V = DP / 4;
    for (int idx = threadIdx.x; idx < 16 * V; idx += kLstmThreads) {
        const int r = idx / V, c4 = idx - r * V;
        if (first_row + r < rows) {
            const float2 *sp = reinterpret_cast<const float2 *>(xs + r * XS + 4 * c4);
            const float2 lo2 = sp[0], hi2 = sp[1];
            *reinterpret_cast<float4 *>(dst + (size_t)(first_row + r) * row_stride + 4 * c4) = make_float4(lo2.x, lo2.y, hi2.x, hi2.y);
        }
    }
}

}  // namespace pfa
