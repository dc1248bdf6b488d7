// ppo_bf16.hpp — the fused forward + PPO loss + backward step of ppo_update.hip (clean_pufferl.py:175-244 up to loss.backward(), for
// models.Default(hidden 128) on 64-float rows) with every contraction on the bf16 matrix path: each fp32 operand is split into three
// bf16 pieces (x = hi + mid + lo, round to nearest each time: 24 mantissa bits in all) and each product is issued as its six partial
// products above 2^-24 relative (lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi — small terms first) on v_mfma_f32_16x16x32_bf16 with
// fp32 accumulation.  OPT-IN (pfa_igemm_set_products(1), bench.py --products bf16x6): as close to the exact product as the fp32 fma
// chain, but not its bit pattern — the default stays the exact-fp32 kernel of ppo_update.hip.
//
// Why a different kernel and not a variant of that one (profiles/r04_ubench_bf16_share.txt): on this part an fp32 MFMA and VALU work
// never overlap on a SIMD (time = sum), a bf16 MFMA stream and another wave's VALU work do (bf16 16x16x32 | VALU fma: 1130 us against
// 255 + 1099), and a 16x16x32 bf16 MFMA takes ~17 cycles against the 32 of a 16x16x4 fp32 one for 8x the products.  So here the
// matrix work hides under the VALU work (loss + operand splitting) of the other workgroup on the CU, and the structure is chosen to
// keep the VALU work small and evenly spread instead of keeping the matrix pipe fed:
//   * a workgroup is FOUR waves working on the same 32-row tile (32 = the contraction depth of the weight gradients, whose
//     contraction index is the batch row); the hidden units are split over the waves (32 each), so a wave's slices of W1 / W2v and of
//     every gradient live in its registers for the whole launch — no fragment tables, two workgroups per CU (<= 256 registers,
//     72.5 KB of LDS);
//   * forward:  hidden^T[u][row] = W1 . X^T with X as three bf16 planes in LDS (split once by the threads that stage the tile);
//     a wave's hidden^T C fragments ARE the B fragments of its K = 32 slice of the heads (a K-permutation shared with the W2v
//     fragment), partial out^T tiles meet in LDS;
//   * loss: the 16 x 16 out^T tile of each 16-row half by ONE wave (two of the four per tile, alternating), d loss / d out as bf16
//     planes [row][o] in LDS;
//   * backward: dh^T = W2v^T . dout^T (B fragments straight from the dout planes), relu' on the C fragment; the three products that
//     contract over batch rows (dW2v^T = hidden^T . dout, dW1^T = X^T . dh) take their operands from ROW-MAJOR bf16 planes through
//     ds_read_b64_tr_b16 (the LDS transposes: tools/experiments/ds_read_tr_map.hip has the lane map) — hidden and dh through the
//     wave's own patch, X and dout from the shared planes.  No operand is transposed by VALU code.
// The partial a workgroup leaves has the layout of ppo_mlp_grad_kernel<64, 0, 13, false, 3, true, PERM> (NativeLayout<64, 3, true>),
// so ppo_reduce_kernel / ppo_reduce_adam_kernel and everything behind them are shared.
#pragma once
#include "common.hpp"
#include "mlp_tile.hpp"
#include "ppo_tile.hpp"

namespace pfa {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

constexpr int kBfThreads = 256;
constexpr int kBfMaxGrid = 512;   // two workgroups per CU

__host__ __device__ constexpr int bf_slot_output(bool perm, int slot) {   // = slot_output of ppo_update.hip
    return !perm ? slot : ((slot & 3) == 3 ? 99 : 3 * (slot >> 2) + (slot & 3));
}

struct BfLds {   // bytes
    static constexpr int XRS = 144;                 // row of an X plane: 64 bf16 + 16 bytes (16-byte reads and transposed reads conflict-free)
    static constexpr int kXPlane = 32 * XRS;
    static constexpr int kXSlot = 3 * kXPlane;      // [piece][row][k]
    static constexpr int DRS = 80;                  // row of a dout plane: 16 slots + 16 zero slots (the K = 32 padding of dh) + 16 bytes
    static constexpr int kDPlane = 32 * DRS;
    static constexpr int HRS = 72;                  // row of a wave's hidden / dh patch plane: 32 units + 8 bytes
    static constexpr int kHPlane = 32 * HRS;
    static constexpr int kX = 0;                                 // [2 slots]
    static constexpr int kPart = kX + 2 * kXSlot;                // [4 waves][2 halves][64 lanes][4] floats: partial out^T tiles
    static constexpr int kD = kPart + 4 * 2 * 64 * 4 * 4;        // [3 pieces] dout planes
    static constexpr int kH = kD + 3 * kDPlane;                  // [4 waves][3 pieces] patches
    static constexpr int kRed = kH + 4 * 3 * kHPlane;            // [4 waves][32] floats: db2 / loss sums at the end
    static constexpr int kBytes = kRed + 4 * 32 * 4;
};

__device__ __forceinline__ uint32_t bf_pk(float a, float b) {   // two floats -> two bf16 (round to nearest even), a in the low half
    f32x2_t v = {a, b};
    bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    return *reinterpret_cast<uint32_t *>(&r);
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __uint_as_float(p & 0xFFFF0000u); }

// x = hi + mid + lo for a pair of values: one packed word per piece
__device__ __forceinline__ void bf_split2(float a, float b, uint32_t &h, uint32_t &m, uint32_t &l) {
    h = bf_pk(a, b);
    const float ra = a - bf_lo(h), rb = b - bf_hi(h);
    m = bf_pk(ra, rb);
    l = bf_pk(ra - bf_lo(m), rb - bf_hi(m));
}
union BfFrag {   // eight bf16 = one operand fragment of v_mfma_f32_16x16x32_bf16: element e sits in word e / 2, half e % 2
    uint32_t w[4];
    bf16x8 v;
    s16x4_t h[2];
};
struct BfFrag3 {
    BfFrag p[3];   // hi, mid, lo
};
__device__ __forceinline__ void bf_split8(const float *x, BfFrag3 &out) {
#pragma unroll
    for (int q = 0; q < 4; ++q) bf_split2(x[2 * q], x[2 * q + 1], out.p[0].w[q], out.p[1].w[q], out.p[2].w[q]);
}
// acc += A . B as the six partial products, small terms first
__device__ __forceinline__ f32x4 bf_mfma(const BfFrag &a, const BfFrag &b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0); }
constexpr int kBfTA[6] = {2, 0, 1, 1, 0, 0}, kBfTB[6] = {0, 2, 1, 0, 1, 0};

// One operand fragment whose contraction index is the ROW of a row-major bf16 plane: element e <-> (row row0 + 8 g + e, column
// col0 + c).  ds_read_b64_tr_b16: within a group of 16 lanes, lane 4 a + b receives element b of the 8 bytes lanes a, a + 4, a + 8,
// a + 12 point at — so lane j of the group points at (row (j >> 2), columns 4 (j & 3) .. + 3) of a 4 x 16 block; two reads = 8 rows.
typedef __attribute__((address_space(3))) s16x4_t *bf_lds_s16x4;
__device__ __forceinline__ void bf_tr8(const unsigned char *plane, int row_stride, int row0, int col0, int c, int g, BfFrag &f) {
    const unsigned char *p = plane + (row0 + 8 * g + (c >> 2)) * row_stride + (col0 + 4 * (c & 3)) * 2;
    f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf_lds_s16x4)p);
    f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf_lds_s16x4)(p + 4 * row_stride));
}

#ifdef PFA_BF16_TRACE
// tools/bf16_trace.py: s_memtime stamps of two workgroups (lane 0 of every wave), [wg slot][wave][tile][16]
__device__ unsigned long long *g_bf_trace = nullptr;
__device__ int g_bf_trace_tiles = 0;
#define BF_STAMP(k)                                                                                                      \
    do {                                                                                                                 \
        if (g_bf_trace && (blockIdx.x == 0 || blockIdx.x == 256) && lane == 0 && j < g_bf_trace_tiles)                   \
            g_bf_trace[(((size_t)(blockIdx.x ? 1 : 0) * 4 + wv) * g_bf_trace_tiles + j) * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define BF_STAMP(k) do { } while (0)
#endif

template <bool PERM>
__global__ void __launch_bounds__(kBfThreads, 2)
    ppo_mlp_grad_bf16_kernel(pfa_experience ex, RowMap map, long long mb_rows, const float *params, int a, pfa_ppo_hparams hp,
                             const double *adv_stats /* [nmb][2] */, double global_rows, float *partials) {
    using L = BfLds;
    constexpr int DP = 64;
    using NL = NativeLayout<DP, 3, true>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id(), wv = wave_id(), c = lane & 15, g = lane >> 4;
    const MlpOffsets off = mlp_offsets(DP, a);
    unsigned char *dpl = smem + L::kD, *patch = smem + L::kH + wv * 3 * L::kHPlane;
    float *part = reinterpret_cast<float *>(smem + L::kPart);

    // ---- this wave's slice of the policy (hidden units 32 wv .. 32 wv + 31 = hidden tiles m = 2 wv + i), split once per launch ------
    BfFrag3 wA[2][2];   // [i][ks]  A of the forward: W1[32wv + 16i + c][32ks + 8g + e]
    BfFrag3 wH;         //          A of the heads:   W2v[slot c][u(e)],  u(e) = 32wv + 16 (e >> 2) + 4g + (e & 3): the C-fragment order of hidden^T
    BfFrag3 wT[2];      // [i]      A of dh:          W2v[slot 8g + e][32wv + 16i + c] for g < 2, zero for g >= 2 (K = 16 slots padded to 32)
    f32x4 hb[2];        // [i]      encoder bias in C-fragment order
    float bo[4];
    {
        float t[8];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const float4 *src = reinterpret_cast<const float4 *>(params + off.w1 + (size_t)(32 * wv + 16 * i + c) * DP + 32 * ks + 8 * g);
                const float4 v0 = src[0], v1 = src[1];
                t[0] = v0.x; t[1] = v0.y; t[2] = v0.z; t[3] = v0.w; t[4] = v1.x; t[5] = v1.y; t[6] = v1.z; t[7] = v1.w;
                bf_split8(t, wA[i][ks]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = g < 2 ? w2v_at(params, off, a, bf_slot_output(PERM, 8 * g + e), 32 * wv + 16 * i + c) : 0.0f;
            bf_split8(t, wT[i]);
#pragma unroll
            for (int r = 0; r < 4; ++r) hb[i][r] = params[off.b1 + 32 * wv + 16 * i + 4 * g + r];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = w2v_at(params, off, a, bf_slot_output(PERM, c), 32 * wv + 16 * (e >> 2) + 4 * g + (e & 3));
        bf_split8(t, wH);
#pragma unroll
        for (int r = 0; r < 4; ++r) bo[r] = b2v_at(params, off, a, bf_slot_output(PERM, 4 * g + r));
    }
    // the K-padding half of the dout planes (slots 16 .. 31) is read by dh's B fragments and never written: zero it once
    for (int i = threadIdx.x; i < 3 * L::kDPlane / 4; i += kBfThreads) reinterpret_cast<uint32_t *>(dpl)[i] = 0u;

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
    const float inv_rows = (float)(1.0 / global_rows), adv_rden = 1.0f / adv_den;

    f32x4 acc_dw1[4][2], acc_dw2[2];   // [k-tile][i]: dW1^T[16kt + 4g + r][32wv + 16i + c];  [i]: dW2v^T[32wv + 16i + 4g + r][slot c]
    float db1[2][4], db2[4], stats[6];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) acc_dw1[kt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc_dw2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) db1[i][r] = 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) db2[r] = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) stats[i] = 0.0f;

    const long long tiles = mb_rows / 32;   // 32-row tiles = pairs of the 16-row tiles RowMap speaks of
    const int J = (int)((tiles + gridDim.x - 1) / gridDim.x);   // the same for every workgroup: everybody runs the same barriers
    const bool aligned = (map.horizon & 15) == 0;
    const int my_half = wv & 1;   // the 16-row half whose loss this wave computes when it is its turn

    // register prefetch of the next tile: two float4 of X per thread (row = idx >> 4, column quad = idx & 15, idx = tid + 256 q: the
    // rows of one q lie in one 16-row half for a whole wave) + the per-row scalars of this wave's half
    float4 xpre[2];
    RowScalars rspre;
    auto prefetch = [&](long long tile) {
        const bool ok = tile < tiles;
        rspre = RowScalars{0, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q) xpre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            {
                const unsigned t16 = 2u * (unsigned)tile + (unsigned)my_half;
                const unsigned first = map.tile_first(t16);
                const unsigned fr = map.tile_row(t16, first, c, aligned);
                rspre = RowScalars{ex.actions[fr], ex.logprobs[fr], ex.values[fr], ex.advantages[fr], ex.returns[fr], 1.0f};
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int idx = threadIdx.x + kBfThreads * q, r = (idx >> 4) & 15, c4 = idx & 15;
                const unsigned t16 = 2u * (unsigned)tile + (unsigned)q;
                const unsigned first = map.tile_first(t16);
                const unsigned row = map.tile_row(t16, first, r, aligned);
                xpre[q] = *reinterpret_cast<const float4 *>(ex.obs + (size_t)row * DP + 4 * c4);
            }
        }
    };
    prefetch(blockIdx.x);

    for (int j = 0; j < J; ++j) {
        unsigned char *xs = smem + L::kX + (j & 1) * L::kXSlot;
        BF_STAMP(0);
        // ---- stage X(j): split into the three planes, 8 bytes (four bf16) per piece and thread ----------------------------------
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = threadIdx.x + kBfThreads * q, row = idx >> 4, c4 = idx & 15;
            uint2 h, m, l;
            bf_split2(xpre[q].x, xpre[q].y, h.x, m.x, l.x);
            bf_split2(xpre[q].z, xpre[q].w, h.y, m.y, l.y);
            unsigned char *d = xs + row * L::XRS + c4 * 8;
            *reinterpret_cast<uint2 *>(d) = h;
            *reinterpret_cast<uint2 *>(d + L::kXPlane) = m;
            *reinterpret_cast<uint2 *>(d + 2 * L::kXPlane) = l;
        }
        const RowScalars rs = rspre;
        prefetch((long long)blockIdx.x + (long long)(j + 1) * gridDim.x);
        BF_STAMP(1);
        __syncthreads();   // A: X(j) visible; every wave is done with tile j - 1 (partials, dout planes, the other X slot)
        BF_STAMP(2);

        // ---- forward slice: hidden^T[32wv + 16i + 4g + r][row 16nt + c] ----------------------------------------------------------
        f32x4 h[2][2];   // [nt][i]
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 2; ++i) h[nt][i] = hb[i];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            BfFrag xb[2][3];   // [nt][piece]: B[k = 32ks + 8g + e][n = row 16nt + c]
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    xb[nt][p].v = *reinterpret_cast<const bf16x8 *>(xs + p * L::kXPlane + (16 * nt + c) * L::XRS + (32 * ks + 8 * g) * 2);
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int i = 0; i < 2; ++i) h[nt][i] = bf_mfma(wA[i][ks].p[kBfTA[t]], xb[nt][kBfTB[t]], h[nt][i]);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) h[nt][i][r] = fmaxf(h[nt][i][r], 0.0f);   // ReLU (models.py:52)
        if (h[0][0][0] == 12345.678f) BF_STAMP(15);   // (never true) pins the stamp behind the forward's result
        BF_STAMP(3);
        // ---- this slice's part of the heads; hidden pieces into the wave's patch [row][unit] for the transposed reads below --------
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float t8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t8[e] = h[nt][e >> 2][e & 3];
            BfFrag3 hp3;
            bf_split8(t8, hp3);
            f32x4 po = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 6; ++t) po = bf_mfma(wH.p[kBfTA[t]], hp3.p[kBfTB[t]], po);
            *reinterpret_cast<f32x4 *>(part + ((wv * 2 + nt) * 64 + lane) * 4) = po;   // out^T[slot 4g + r][row 16nt + c], this slice's share
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i)   // elements 4i .. 4i + 3 = units 16i + 4g .. + 3 of row 16nt + c
                    *reinterpret_cast<uint2 *>(patch + p * L::kHPlane + (16 * nt + c) * L::HRS + (16 * i + 4 * g) * 2) =
                        make_uint2(hp3.p[p].w[2 * i], hp3.p[p].w[2 * i + 1]);
        }
        BF_STAMP(4);
        __syncthreads();   // B: the four partials of out^T
        BF_STAMP(5);

        // ---- loss of one 16-row half per wave, two waves per tile (the pair alternates from tile to tile) ---------------------------
        if ((wv >> 1) == (j & 1)) {
            const int nt = my_half;
            f32x4 out;
            {
                const f32x4 p0 = *reinterpret_cast<const f32x4 *>(part + ((0 * 2 + nt) * 64 + lane) * 4);
                const f32x4 p1 = *reinterpret_cast<const f32x4 *>(part + ((1 * 2 + nt) * 64 + lane) * 4);
                const f32x4 p2 = *reinterpret_cast<const f32x4 *>(part + ((2 * 2 + nt) * 64 + lane) * 4);
                const f32x4 p3 = *reinterpret_cast<const f32x4 *>(part + ((3 * 2 + nt) * 64 + lane) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) out[r] = bo[r] + ((p0[r] + p1[r]) + (p2[r] + p3[r]));
            }
            const LossOut lo = ppo_loss_tile<false, true, PERM>(out, rs, a, 0u, g, hp, adv_mean, adv_rden, inv_rows);
#pragma unroll
            for (int r = 0; r < 4; ++r) db2[r] += lo.dout[r];
            if (g == 0) {
                stats[0] += lo.pg;
                stats[1] += lo.v_loss;
                stats[2] += lo.ent;
                stats[3] += lo.neg_logratio;
                stats[4] += lo.kl;
                stats[5] += lo.clipped;
            }
            uint2 dh_, dm_, dl_;
            bf_split2(lo.dout[0], lo.dout[1], dh_.x, dm_.x, dl_.x);
            bf_split2(lo.dout[2], lo.dout[3], dh_.y, dm_.y, dl_.y);
            unsigned char *d = dpl + (16 * nt + c) * L::DRS + 8 * g;   // dout[row 16nt + c][slots 4g .. 4g + 3]
            *reinterpret_cast<uint2 *>(d) = dh_;
            *reinterpret_cast<uint2 *>(d + L::kDPlane) = dm_;
            *reinterpret_cast<uint2 *>(d + 2 * L::kDPlane) = dl_;
        }
        if (db2[0] == 12345.678f) BF_STAMP(15);
        BF_STAMP(6);
        __syncthreads();   // C: d loss / d out of both halves
        BF_STAMP(7);

        // ---- backward slice -----------------------------------------------------------------------------------------------------
        // dW2v^T[u][slot] += sum over the 32 rows of hidden[row][u] dout[row][slot]: both operands by transposed reads
        {
            BfFrag db_[3], ha[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                bf_tr8(dpl + p * L::kDPlane, L::DRS, 0, 0, c, g, db_[p]);
#pragma unroll
                for (int i = 0; i < 2; ++i) bf_tr8(patch + p * L::kHPlane, L::HRS, 0, 16 * i, c, g, ha[i][p]);
            }
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc_dw2[i] = bf_mfma(ha[i][kBfTA[t]], db_[kBfTB[t]], acc_dw2[i]);
        }
        // dh^T[u][row] = sum over the slots of W2v[slot][u] dout[row][slot]; relu' from the forward's own C fragments; db1
        f32x4 dh[2][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            BfFrag dB[3];   // B[k = slot 8g + e][n = row 16nt + c]; slots 16 .. 31 are the zero half
#pragma unroll
            for (int p = 0; p < 3; ++p) dB[p].v = *reinterpret_cast<const bf16x8 *>(dpl + p * L::kDPlane + (16 * nt + c) * L::DRS + 16 * g);
#pragma unroll
            for (int i = 0; i < 2; ++i) dh[nt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) dh[nt][i] = bf_mfma(wT[i].p[kBfTA[t]], dB[kBfTB[t]], dh[nt][i]);
        }
        if (dh[0][0][0] == 12345.678f || acc_dw2[0][0] == 12345.678f) BF_STAMP(15);
        BF_STAMP(8);
        wave_lds_fence();   // this wave's transposed reads of the hidden pieces are issued before the patch is overwritten (LDS is in order)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float t8[8];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = h[nt][i][r] > 0.0f ? dh[nt][i][r] : 0.0f;   // relu'
                    db1[i][r] += v;
                    t8[4 * i + r] = v;
                }
            BfFrag3 d3;
            bf_split8(t8, d3);
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    *reinterpret_cast<uint2 *>(patch + p * L::kHPlane + (16 * nt + c) * L::HRS + (16 * i + 4 * g) * 2) =
                        make_uint2(d3.p[p].w[2 * i], d3.p[p].w[2 * i + 1]);
        }
        wave_lds_fence();
        BF_STAMP(9);
        // dW1^T[k][u] += sum over the 32 rows of X[row][k] dh[row][u]
        {
            BfFrag dhb[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i) bf_tr8(patch + p * L::kHPlane, L::HRS, 0, 16 * i, c, g, dhb[i][p]);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                BfFrag xa[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) bf_tr8(xs + p * L::kXPlane, L::XRS, 0, 16 * kt, c, g, xa[p]);
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc_dw1[kt][i] = bf_mfma(xa[kBfTA[t]], dhb[i][kBfTB[t]], acc_dw1[kt][i]);
            }
        }
        if (acc_dw1[0][0][0] == 12345.678f) BF_STAMP(15);
        BF_STAMP(10);
        wave_lds_fence();   // ... and the reads of the dh pieces before the next tile's hidden pieces land in the patch
    }

    // ---- this workgroup's partial: every wave its own slices, in the fragment order of NativeLayout<64, 3, true> --------------------
    float *dst = partials + (size_t)blockIdx.x * NL::kCount;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = 2 * wv + i;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[NL::kDw1 + ((kt * kMT + m) * 4 + r) * 64 + lane] = acc_dw1[kt][i][r];
        if (g == 0) dst[NL::kCol + 16 * m + c] = acc_dw1[3][i][0];   // column 48 = row (g = 0, r = 0) of k-tile 3
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[NL::kDw2 + (m * 4 + r) * 64 + lane] = acc_dw2[i][r];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float b = db1[i][r];
#pragma unroll
            for (int s = 1; s < 16; s <<= 1) b += __shfl_xor(b, s, 64);
            if (c == 0) dst[NL::kDb1 + 16 * m + 4 * g + r] = b;
        }
    }
    float *red = reinterpret_cast<float *>(smem + L::kRed) + wv * 32;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int s = 1; s < 16; s <<= 1) db2[r] += __shfl_xor(db2[r], s, 64);
        if (c == 0) red[4 * g + r] = db2[r];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int s = 1; s < 16; s <<= 1) stats[i] += __shfl_xor(stats[i], s, 64);
        if (lane == 0) red[16 + i] = stats[i];
    }
    __syncthreads();
    if (wv == 0 && lane < 16 + kNumStats) {
        const float *r0 = reinterpret_cast<const float *>(smem + L::kRed);
        const bool live = lane < 16 + 6;
        const float s = live ? (r0[lane] + r0[32 + lane]) + (r0[64 + lane] + r0[96 + lane]) : 0.0f;
        dst[(lane < 16 ? NL::kDb2 : NL::kStats - 16) + lane] = s;
    }
}

}  // namespace pfa
