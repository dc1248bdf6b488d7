// ppo_wide.hip — the minibatch step of clean_pufferl.train (clean_pufferl.py:175-244 up to loss.backward()) for a Default policy of
// another width than the 128 the kernel of ppo_update.hip is built around: models.Default(env, hidden_size=H) (models.py:24-62)
// with H in {64, 256, 512} — what environments/classic_control, nethack / nmmo and atari's MLP use — one Discrete head of up to 15
// actions, observation rows of 16 / 32 / 64 floats.  One launch = forward + sample_logits with given actions (cleanrl.py:25-47) +
// PPO loss (clean_pufferl.py:202-238) + backward over one minibatch, every contraction on v_mfma_f32_16x16x4_f32 (exact fp32
// chains: the 1e-5 parity target), the hidden layer never leaving registers / LDS.
//
// Structure: the hidden units are independent of each other until the heads sum over them, so the HIDDEN dimension is split over
// the four wavefronts of a workgroup — wave w owns units [H/4 w, H/4 (w+1)) for the whole launch: its W1 rows, its columns of
// W2 / Wv and its slices of every gradient live in ITS registers for the whole launch (MS = H / 64 hidden tiles per wave: KKU MS
// W1 fragments + 4 MS head fragments + (4 KTM + 4) MS gradient accumulators, ~300 of the 512 registers a one-wave-per-SIMD
// workgroup has at H = 512; the B fragments of dh and the bias sit in LDS).  All four waves work on the SAME 16-row tile:
//   all      X tile -> LDS (next tile's rows prefetched into registers a tile ahead)            | barrier
//   wave w   hidden^T slice = W1_w . X^T  (MS x KKU MFMAs), ReLU, partial out^T = W2v_w . hidden_w  -> LDS                | barrier
//   wave w   out^T = bias + sum of the four partials (fixed order); the PPO loss of the 16 rows and d loss / d out — computed by
//            every wave for itself (it needs all of dout, and the loss is cheaper than another barrier and a broadcast)
//            dh_w = dout . W2v_w, relu' from its own hidden slice (transposed through its own LDS patch: no workgroup barrier),
//            dW2v_w += hidden_w^T dout,  dW1_w += X^T dh_w,  db1_w += column sums of dh_w
// Two workgroup barriers per tile; the X tile is double-buffered so that the next tile's rows can be stored while a slow wave still
// reads this one.  At the end every wave writes its gradient slices (MFMA C-fragment order, coalesced) as this workgroup's
// partial; ppo_wide_reduce_kernel sums the partials in a fixed order and scatters into torch's tensor shapes through a second
// view (the gradient buffer of general.GeneralParams: named_parameters() order).  Clip + Adam: pfa_adam_clip_step on the flat buffer.
#include <cmath>

#include "common.hpp"
#include "mlp_tile.hpp"
#include "ppo_tile.hpp"
#include "rollout_tile.hpp"

namespace pfa {

#ifndef PFA_WIDE_OCC
#define PFA_WIDE_OCC 1   // 0 = one workgroup per CU for every width (the first form of this kernel; A/B timing with tools/wide_variant_bench.py)
#endif
// Workgroups per CU.  A workgroup is one wave per SIMD, all four on the same tile behind two barriers per tile: nothing of its own
// covers a wave's LDS round trips, barrier waits and the (VALU-only) loss.  Where the register budget allows, a second (third,
// fourth) workgroup on the CU does: MS = 1 fits four (<= 128 registers), MS = 2 / 4 two (<= 256; at MS = 4 with the heads' A
// fragments in LDS instead of registers and two hidden tiles per backward pass instead of four); MS = 8 needs the whole file.
// (KKU = k-steps of the forward: MS KKU W1 fragments + 4 MS KTM dW1 accumulators set the budget; 64 full columns at MS = 4 do not fit twice.)
__host__ __device__ constexpr int wide_wgs_per_cu(int ms, int kku) { return !PFA_WIDE_OCC ? 1 : ms == 1 ? 4 : (ms <= 4 && ms * kku <= 52) ? 2 : 1; }
#ifndef PFA_WIDE_8W
#define PFA_WIDE_8W 1    // 0 = hidden 512 as four waves of MS = 8 (the first form; A/B timing)
#endif
#ifndef PFA_WIDE_DEDUP
#define PFA_WIDE_DEDUP 1 // 0 = in the eight-wave form every wave computes the loss for itself (A/B timing)
#endif
// Waves per workgroup (NW) and hidden tiles per wave (MS = hidden / (16 NW)).  Hidden 512 at four waves needs the whole register file
// (MS = 8: 459-498 registers), so nothing covers its latencies either; as EIGHT waves of MS = 4 — two per SIMD, <= 256 registers each,
// the same budget as hidden 256 with two workgroups per CU — each SIMD has a second wave to issue from.  The two waves of a SIMD would
// both spend the loss's VALU time (MFMA and VALU do not overlap on a SIMD: DESIGN.md 3.4), so only waves 0-3 compute it and hand
// d loss / d out to their SIMD partner (wave w + 4) through LDS behind a third barrier (DEDUP).
__host__ __device__ constexpr int wide_waves(int hidden, int kku) { return PFA_WIDE_8W && hidden == 512 && 4 * kku <= 52 ? 8 : 4; }
__host__ __device__ constexpr int wide_waves_per_simd(int ms, int kku, int nw) { return nw == 8 ? 2 : wide_wgs_per_cu(ms, kku); }
__host__ __device__ constexpr bool wide_heads_in_lds(int ms, int kku, int nw) { return ms == 4 && wide_waves_per_simd(ms, kku, nw) > 1; }
__host__ __device__ constexpr bool wide_small_pass(int ms, int kku, int nw) { return wide_waves_per_simd(ms, kku, nw) > 1; }
#ifndef PFA_WIDE_STREAMS
#define PFA_WIDE_STREAMS 1   // 0 = the co-resident tile streams of a CU as separate workgroups, each with its own gradient partial (round 5)
#endif
// Tile STREAMS per workgroup.  Where a CU holds two (four) four-wave groups, they used to be two (four) workgroups — and every
// workgroup leaves its own gradient partial: 512 x 67.6 KB at hidden 256 = 34.6 MB written per launch and read again by the reduce
// launch, as much as the minibatch's rows themselves (2.07x the algorithmic traffic by the counters, VERDICT rounds 4 and 5).  As ONE
// workgroup of ST four-wave streams — each stream its own tiles, X / hidden / dout patches and partial out^T, all streams behind
// the same two barriers per tile — the streams add their accumulators up through LDS at the end (fixed order) and the workgroup
// writes ONE partial: half (a quarter) of the partial traffic in both launches.
__host__ __device__ constexpr int wide_streams(int ms, int kku, int nw) { return PFA_WIDE_STREAMS && nw == 4 ? wide_wgs_per_cu(ms, kku) : 1; }

// One workgroup partial (floats), fragment order.  MT = H / 16 hidden tiles, KTM 16-column tiles of dW1 on MFMA (+ COL: the
// one trailing column 16 KTM, for rows of 16 KTM + 1 real columns like the 7x7 grid's 49).
struct WideLayout {
    int MT, KTM, col;
    int kCol, kDw2, kDb1, kDb2, kStats, kCount;
};
__host__ __device__ inline WideLayout wide_layout(int hidden, int ktm, bool col) {
    WideLayout l;
    l.MT = hidden / 16;
    l.KTM = ktm;
    l.col = col ? 1 : 0;
    l.kCol = ktm * l.MT * 256;                 // ((kt*MT + m)*4 + r)*64 + lane -> W1[16m + c][16kt + 4g + r]
    l.kDw2 = l.kCol + (col ? hidden : 0);      // u = 16m + c -> W1[u][16 KTM]
    l.kDb1 = l.kDw2 + l.MT * 256;              // (m*4 + r)*64 + lane -> W2v[o = c][16m + 4g + r]
    l.kDb2 = l.kDb1 + hidden;
    l.kStats = l.kDb2 + kOut;
    l.kCount = l.kStats + kNumStats;
    return l;
}

// LDS map (floats): two X tiles, the four partial out^T tiles, one hidden-slice patch and one dout patch per wave, the W2v
// B-fragment table and the encoder bias.
template <int DP, int MS, int KKU, int NW = 4, int KTM = 4>
struct WideLds {
    static constexpr int ST = wide_streams(MS, KKU, NW);
    static constexpr int XS = XTile<DP>::XS, HSW = 16 * MS + 4, DSW = 20;
    // per stream: [ST][kStream]
    static constexpr int kXs = 0;                          // [2][16 * XS]
    static constexpr int kPart = kXs + 2 * 16 * XS;        // [NW][16 * 16]
    static constexpr int kHs = kPart + NW * kOut * 16;     // [NW][16 * HSW]
    static constexpr int kDs = kHs + NW * 16 * HSW;        // [NW][16 * DSW]
    static constexpr int kStream = kDs + NW * 16 * DSW;
    // shared by the streams (stream 0 writes them before the first barrier)
    static constexpr int kWb = ST * kStream;               // [NW MS][64][4]
    static constexpr int kB1 = kWb + NW * MS * 256;        // [16 NW MS]
    static constexpr int kWa = kB1 + 16 * NW * MS;         // [NW MS][64][4]: the heads' A fragments (wide_heads_in_lds)
    static constexpr int kTiles = kWa + (wide_heads_in_lds(MS, KKU, NW) ? NW * MS * 256 : 0);
    // after the tile loop the same memory carries the accumulators of streams 1 .. ST-1 to stream 0: [ST - 1][NW][kAcc][64]
    static constexpr int kAcc = 4 * KTM * MS + 4 * MS + 2 * MS + 4 + 6;
    static constexpr int kHand = (ST - 1) * NW * kAcc * 64;
    static constexpr int kFloats = kTiles > kHand ? kTiles : kHand;
};

template <int DP, int KKU, int KTM, bool COL, int MS, int NW = 4>
__global__ void __launch_bounds__(64 * NW * wide_streams(MS, KKU, NW), wide_waves_per_simd(MS, KKU, NW)) ppo_wide_grad_kernel(pfa_experience ex, RowMap map, long long mb_rows, MlpView pv, pfa_ppo_hparams hp,
                                                                    const double *adv_stats /* [nmb][2] */, double global_rows, float *partials) {
    using LD = WideLds<DP, MS, KKU, NW, KTM>;
    constexpr int XS = LD::XS, V = DP / 4, HSW = LD::HSW, DSW = LD::DSW, MT = NW * MS, ST = LD::ST;
    constexpr bool DEDUP = PFA_WIDE_DEDUP && NW == 8;
    constexpr bool LATE_PREFETCH = NW == 8;
    static_assert(NW == 4 || NW == 8, "waves per workgroup");
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    // stream st = four-wave group of this workgroup (uniform per wave); wv = the wave's hidden slice within its stream
    const int st = ST > 1 ? wave_id() / NW : 0, wv = ST > 1 ? wave_id() % NW : wave_id();
    const int tl = (int)threadIdx.x - st * 64 * NW;     // thread index within the stream
    float *lds = lds_all + st * LD::kStream;            // this stream's tiles; the fragment tables below are shared
    float *part = lds + LD::kPart;             // [NW][16 * 16] partial out^T per wave
    const int lane = lane_id(), c = lane & 15, g = lane >> 4;
    const int a = pv.a;
    // (eight-wave form with DEDUP: wave w + 4 reads the dout patch its SIMD partner w wrote)
    float *hs = lds + LD::kHs + wv * 16 * HSW, *ds = lds + LD::kDs + ((PFA_WIDE_DEDUP && NW == 8) ? (wv & 3) : wv) * 16 * DSW;
    float *wbt = lds_all + LD::kWb, *b1s = lds_all + LD::kB1;
    constexpr bool W2L = wide_heads_in_lds(MS, KKU, NW);
    float *wat = lds_all + LD::kWa;

    // ---- this wave's slice of the policy, as MFMA fragments, for the whole launch: W1 and the heads' A fragments in registers, the
    // B fragments of dh = dout . W2v and the encoder bias in LDS (written and read by this wave only) ------------------------------
    float w1f[MS][KKU];
    f32x4 w2f[W2L ? 1 : MS];
    float bo[4];
#pragma unroll
    for (int i = 0; i < MS; ++i) {
        const int m = MS * wv + i;
#pragma unroll
        for (int kk = 0; kk < KKU; ++kk) w1f[i][kk] = 4 * kk + g < pv.cols ? pv.w1[(size_t)(16 * m + c) * pv.ldw1 + 4 * kk + g] : 0.0f;
        f32x4 wb, wa;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            wa[r] = pv.w2v(c, 16 * m + 4 * g + r);            // A[i = o = c][k-slot g] of the heads, u = 16m + 4g + r
            wb[r] = pv.w2v(4 * g + r, 16 * m + c);            // B[k-slot g][j = u = 16m + c] of dh = dout . W2v, o = 4g + r
        }
        if constexpr (W2L) {
            if (st == 0) *reinterpret_cast<f32x4 *>(wat + (m * 64 + lane) * 4) = wa;
        } else {
            w2f[i] = wa;
        }
        if (st == 0) {   // (the other streams read the tables behind the loop's first barrier)
            *reinterpret_cast<f32x4 *>(wbt + (m * 64 + lane) * 4) = wb;
            if (lane < 16) b1s[16 * m + lane] = pv.b1[16 * m + lane];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) bo[r] = pv.b2v(4 * g + r);
    wave_lds_fence();

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

    f32x4 acc_dw1[KTM][MS], acc_dw2[MS];
    float acc_col[MS], db1[MS], db2[4], stats[6];
#pragma unroll
    for (int i = 0; i < MS; ++i) {
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) acc_dw1[kt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc_dw2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc_col[i] = db1[i] = 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) db2[r] = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) stats[i] = 0.0f;

    const long long tiles = mb_rows / 16;
    const long long GS = (long long)gridDim.x * ST, gs = (long long)blockIdx.x * ST + st;   // tile streams of the launch, this one
    const int J = (int)((tiles + GS - 1) / GS);   // the same for every stream: everybody runs the same barriers
    const bool aligned = (map.horizon & 15) == 0;

    // register prefetch of the next tile: one float4 of X per thread (threads < 16 V) + the per-row scalars (every wave its own copy)
    float4 xpre = make_float4(0.f, 0.f, 0.f, 0.f);
    RowScalars rspre{0, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto prefetch = [&](long long tile) {
        const bool ok = tile < tiles;
        const unsigned first = ok ? map.tile_first((unsigned)tile) : 0u;
        rspre = RowScalars{0, 0.f, 0.f, 0.f, 0.f, 0.f};
        xpre = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            const unsigned fr = map.tile_row((unsigned)tile, first, c, aligned);
            rspre = RowScalars{ex.actions[fr], ex.logprobs[fr], ex.values[fr], ex.advantages[fr], ex.returns[fr], 1.0f};
            const int idx = tl;
            if (idx < 16 * V) {
                const int r = idx / V, c4 = idx - r * V;
                const unsigned row = map.tile_row((unsigned)tile, first, r, aligned);
                xpre = *reinterpret_cast<const float4 *>(ex.obs + (size_t)row * DP + 4 * c4);
            }
        }
    };
    prefetch(gs);

    for (int j = 0; j < J; ++j) {
        float *xs = lds + LD::kXs + (j & 1) * 16 * XS;
        if (tl < 16 * V) {
            const int r = tl / V, c4 = tl - r * V;
            float2 *d = reinterpret_cast<float2 *>(xs + r * XS + 4 * c4);
            d[0] = make_float2(xpre.x, xpre.y);
            d[1] = make_float2(xpre.z, xpre.w);
        }
        const RowScalars rs = rspre;
        if constexpr (!LATE_PREFETCH) prefetch(gs + (long long)(j + 1) * GS);
        __syncthreads();   // X(j) visible; every wave is done with tile j-1 (its partials and the other X slot)

        // ---- forward slice ---------------------------------------------------------------------------------------------------
        f32x4 h[MS];
#pragma unroll
        for (int i = 0; i < MS; ++i) h[i] = *reinterpret_cast<const f32x4 *>(b1s + 16 * (MS * wv + i) + 4 * g);
#pragma unroll
        for (int kk = 0; kk < KKU; ++kk) {
            const float b = xs[c * XS + 4 * kk + g];
#pragma unroll
            for (int i = 0; i < MS; ++i) h[i] = mfma16(w1f[i][kk], b, h[i]);
        }
#pragma unroll
        for (int i = 0; i < MS; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[i][r] = fmaxf(h[i][r], 0.0f);
        f32x4 o0 = f32x4{0.f, 0.f, 0.f, 0.f}, o1 = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (W2L) {
            f32x4 wa[MS];   // issued ahead of the ReLU's VALU stretch above would be better still; the other workgroup covers the wait
#pragma unroll
            for (int i = 0; i < MS; ++i) wa[i] = *reinterpret_cast<const f32x4 *>(wat + ((MS * wv + i) * 64 + lane) * 4);
#pragma unroll
            for (int i = 0; i < MS; i += 2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o0 = mfma16(wa[i][r], h[i][r], o0);
                    o1 = mfma16(wa[i + 1][r], h[i + 1][r], o1);
                }
        } else {
#pragma unroll
        for (int i = 0; i < MS; i += 2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o0 = mfma16(w2f[i][r], h[i][r], o0);
                if constexpr (MS >= 2) o1 = mfma16(w2f[i + 1][r], h[i + 1][r], o1);
            }
        }
        {
            const f32x4 po = o0 + o1;   // partial out^T[o = 4g + r][row = c] over this wave's hidden units
#pragma unroll
            for (int r = 0; r < 4; ++r) part[wv * (kOut * 16) + (4 * g + r) * 16 + c] = po[r];
        }
        // this wave's hidden slice [row][u] and (below) dout [row][o] go through its own LDS patch: same-wave write -> read
#pragma unroll
        for (int i = 0; i < MS; ++i) *reinterpret_cast<f32x4 *>(hs + c * HSW + 16 * i + 4 * g) = h[i];
        __syncthreads();   // the four partials of out^T

        // ---- out^T, loss, d loss / d out (every wave for itself) --------------------------------------------------------------
        f32x4 dout;
        if (!DEDUP || wv < 4) {
            f32x4 out;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = (4 * g + r) * 16 + c;
                float sum = (part[q] + part[kOut * 16 + q]) + (part[2 * kOut * 16 + q] + part[3 * kOut * 16 + q]);
                if constexpr (NW == 8)
                    sum += (part[4 * kOut * 16 + q] + part[5 * kOut * 16 + q]) + (part[6 * kOut * 16 + q] + part[7 * kOut * 16 + q]);
                out[r] = bo[r] + sum;
            }
            const LossOut lo = ppo_loss_tile<false, true, false>(out, rs, a, 0u, g, hp, adv_mean, adv_rden, inv_rows);
            dout = lo.dout;
            if (wv == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) db2[r] += dout[r];
                if (g == 0) {
                    stats[0] += lo.pg;
                    stats[1] += lo.v_loss;
                    stats[2] += lo.ent;
                    stats[3] += lo.neg_logratio;
                    stats[4] += lo.kl;
                    stats[5] += lo.clipped;
                }
            }
            *reinterpret_cast<f32x4 *>(ds + c * DSW + 4 * g) = dout;   // dout[row = c][o = 4g .. 4g+3]
        }
        // (two waves per SIMD: the next tile's rows and scalars are fetched from here, behind the loss, so that they and this tile's
        // scalars are never live together — the ~2500 cycles of the backward's MFMAs cover the fetch)
        if constexpr (LATE_PREFETCH) prefetch(gs + (long long)(j + 1) * GS);
        if constexpr (DEDUP) {
            __syncthreads();   // d loss / d out of waves 0-3 -> their SIMD partners
            if (wv >= 4) dout = *reinterpret_cast<const f32x4 *>(ds + c * DSW + 4 * g);
        } else {
            wave_lds_fence();
        }

        // ---- backward slice ---------------------------------------------------------------------------------------------------
        float dfrag[4], xa[KTM][4], xc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dfrag[r] = ds[(4 * g + r) * DSW + c];              // B[k-slot g <-> row 4g+r][j = o = c]
#pragma unroll
            for (int kt = 0; kt < KTM; ++kt) xa[kt][r] = xs[(4 * g + r) * XS + 16 * kt + c];
            xc[r] = COL ? xs[(4 * g + r) * XS + 16 * KTM] : 0.0f;
        }
        // hidden tiles per pass: bounds the registers of the short-lived fragments (two where a second workgroup shares the register file)
        constexpr int CH = wide_small_pass(MS, KKU, NW) ? (MS < 2 ? MS : 2) : (MS < 4 ? MS : 4);
#pragma unroll
        for (int i0 = 0; i0 < MS; i0 += CH) {
            float hrow[CH][4];
            f32x4 dh[CH], wb[CH];
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                dh[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                wb[q] = *reinterpret_cast<const f32x4 *>(wbt + ((MS * wv + i0 + q) * 64 + lane) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) hrow[q][r] = hs[(4 * g + r) * HSW + 16 * (i0 + q) + c];   // hidden[row = 4g+r][u = 16m + c]
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < CH; ++q) {
                    acc_dw2[i0 + q] = mfma16(hrow[q][r], dfrag[r], acc_dw2[i0 + q]);     // dW2v^T[u][o] += hidden^T . dout
                    dh[q] = mfma16(dout[r], wb[q][r], dh[q]);                           // dh[row][u] = dout . W2v
                }
#pragma unroll
            for (int q = 0; q < CH; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dh[q][r] = hrow[q][r] > 0.0f ? dh[q][r] : 0.0f;   // relu'
                    db1[i0 + q] += dh[q][r];
                    if (COL) acc_col[i0 + q] = fmaf(xc[r], dh[q][r], acc_col[i0 + q]);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < CH; ++q)
#pragma unroll
                    for (int kt = 0; kt < KTM; ++kt) acc_dw1[kt][i0 + q] = mfma16(xa[kt][r], dh[q][r], acc_dw1[kt][i0 + q]);   // dW1^T[k][u] += X^T . dh
        }
    }

    // ---- streams 1 .. ST-1 hand their accumulators to stream 0 through LDS (the tile buffers are dead), added in stream order ----------
    if constexpr (ST > 1) {
        constexpr int kAcc = LD::kAcc;
        __syncthreads();   // every stream is done with its last tile
        auto slot = [&](int s, int k) -> float & { return lds_all[(((s - 1) * NW + wv) * kAcc + k) * 64 + lane]; };
        auto each = [&](auto &&f) {   // the same walk over the accumulators for the writer and the reader: v <- f(v, k)
            int k = 0;
#pragma unroll
            for (int i = 0; i < MS; ++i) {
#pragma unroll
                for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc_dw1[kt][i][r] = f(acc_dw1[kt][i][r], k++);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc_dw2[i][r] = f(acc_dw2[i][r], k++);
                db1[i] = f(db1[i], k++);
                acc_col[i] = f(acc_col[i], k++);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) db2[r] = f(db2[r], k++);
#pragma unroll
            for (int i = 0; i < 6; ++i) stats[i] = f(stats[i], k++);
        };
        if (st > 0) each([&](float v, int k) { slot(st, k) = v; return v; });
        __syncthreads();
        if (st > 0) return;
#pragma unroll
        for (int s = 1; s < ST; ++s) each([&](float v, int k) { return v + slot(s, k); });
    }
    // ---- this workgroup's partial: every wave its own slices, fragment order ------------------------------------------------------
    const WideLayout L = wide_layout(16 * MT, KTM, COL);
    float *dst = partials + (size_t)blockIdx.x * L.kCount;
#pragma unroll
    for (int i = 0; i < MS; ++i) {
        const int m = MS * wv + i;
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[((kt * MT + m) * 4 + r) * 64 + lane] = acc_dw1[kt][i][r];
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[L.kDw2 + (m * 4 + r) * 64 + lane] = acc_dw2[i][r];
        float b = db1[i];
        b += __shfl_xor(b, 16, 64);
        b += __shfl_xor(b, 32, 64);
        if (g == 0) dst[L.kDb1 + 16 * m + c] = b;
        if (COL) {
            const float s = gsum<true>(acc_col[i]);   // over the lane groups: all 16 rows of the tile
            if (g == 0) dst[L.kCol + 16 * m + c] = s;
        }
    }
    if (wv == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int s = 1; s < 16; s <<= 1) db2[r] += __shfl_xor(db2[r], s, 64);
            if (c == 0) dst[L.kDb2 + 4 * g + r] = db2[r];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int s = 1; s < 16; s <<= 1) stats[i] += __shfl_xor(stats[i], s, 64);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < kNumStats; ++i) dst[L.kStats + i] = i < 6 ? stats[i] : 0.0f;
        }
    }
}

// Fixed-order sum of the workgroup partials + scatter into torch's tensor shapes through the gradient view (same tensors as the
// parameter view, pointing into the gradient buffer); the six loss sums in f64 -> (hi, lo) float pairs in tail16.
constexpr int kWideRedSl = 16;
__global__ void __launch_bounds__(64 * kWideRedSl) ppo_wide_reduce_kernel(const float *partials, int nparts, WideLayout L, int obs_dim, int a,
                                                                         int hidden, float *gw1, int ldw1, float *gb1, float *gw2, float *gb2,
                                                                         float *gwv, float *gbv, float *tail16) {
    __shared__ float sh[kWideRedSl][64];
    __shared__ double shd[kWideRedSl][64];
    const int ql = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + ql;
    float acc = 0.0f;
    double dacc = 0.0;
    if (q < L.kCount) {
        for (int i0 = sl; i0 < nparts; i0 += 16 * kWideRedSl) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = i0 + kWideRedSl * u;
                v[u] = i < nparts ? partials[(size_t)i * L.kCount + q] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc += v[u];
                if (q >= L.kStats) dacc += (double)v[u];
            }
        }
    }
    sh[sl][ql] = acc;
    shd[sl][ql] = dacc;
    __syncthreads();
    if (sl != 0 || q >= L.kCount) return;
    float t[kWideRedSl];
#pragma unroll
    for (int w = 0; w < kWideRedSl; ++w) t[w] = sh[w][ql];
#pragma unroll
    for (int w = kWideRedSl / 2; w > 0; w >>= 1)
#pragma unroll
        for (int x = 0; x < w; ++x) t[x] += t[x + w];
    const float s = t[0];
    if (q < L.kCol) {                         // dW1 on MFMA: W1[16m + c][16kt + 4g + r]
        const int ln = q & 63, r = (q >> 6) & 3, mm = (q >> 8) % L.MT, kt = (q >> 8) / L.MT;
        const int u = 16 * mm + (ln & 15), k = 16 * kt + 4 * (ln >> 4) + r;
        if (k < obs_dim) gw1[(size_t)u * ldw1 + k] = s;
    } else if (q < L.kDw2) {                  // the trailing column
        const int k = 16 * L.KTM;
        if (k < obs_dim) gw1[(size_t)(q - L.kCol) * ldw1 + k] = s;
    } else if (q < L.kDb1) {                  // dW2v: W2v[o = c][16m + 4g + r]
        const int t2 = q - L.kDw2, ln = t2 & 63, r = (t2 >> 6) & 3, mm = t2 >> 8;
        const int o = ln & 15, u = 16 * mm + 4 * (ln >> 4) + r;
        if (o < a) gw2[(size_t)o * hidden + u] = s;
        else if (o == a) gwv[u] = s;
    } else if (q < L.kDb2) {
        gb1[q - L.kDb1] = s;
    } else if (q < L.kStats) {
        const int o = q - L.kDb2;
        if (o < a) gb2[o] = s;
        else if (o == a) gbv[0] = s;
    } else {
        double t2 = 0.0;
#pragma unroll
        for (int w = 0; w < kWideRedSl; ++w) t2 += shd[w][ql];
        const float hi = (float)t2;
        tail16[2 * (q - L.kStats)] = hi;
        tail16[2 * (q - L.kStats) + 1] = (float)(t2 - (double)hi);
    }
}

static int wide_shape(const pfa_mlp_view *p, int *ktm, bool *col, int *kku) {
    PFA_REQUIRE(p && p->w1 && p->b1 && p->w2 && p->b2 && p->wv && p->bv, "ppo.wide: null view / tensor");
    PFA_REQUIRE(p->hidden == 64 || p->hidden == 256 || p->hidden == 512, "ppo.wide: hidden %d is not one of 64 / 256 / 512 (128: pfa_ppo_mlp_grad)", p->hidden);
    PFA_REQUIRE(p->obs_stride == 16 || p->obs_stride == 32 || p->obs_stride == 64, "ppo.wide: obs_stride %d is not one of 16 / 32 / 64", p->obs_stride);
    PFA_REQUIRE(p->obs_dim >= 1 && p->obs_dim <= p->obs_stride && p->ldw1 >= p->obs_dim, "ppo.wide: obs_dim %d / ldw1 %d out of range", p->obs_dim, p->ldw1);
    PFA_REQUIRE(p->num_actions >= 1 && p->num_actions <= 15, "ppo.wide: num_actions must be in 1..15 (got %d)", p->num_actions);
    // the 7x7 grid's 49 columns on a 64-float row: three 16-column tiles on MFMA + the one trailing column on the VALU, 13 k-steps
    const bool grid49 = p->obs_stride == 64 && p->obs_dim == 49;
    *ktm = grid49 ? 3 : p->obs_stride / 16;
    *col = grid49;
    *kku = grid49 ? 13 : p->obs_stride / 4;
    return 0;
}
static int wide_streams_of(int hidden, int kku) {
    const int nw = wide_waves(hidden, kku);
    return wide_streams(hidden / (16 * nw), kku, nw);
}
static int wide_slots(int hidden, int kku) {   // workgroups the chip holds at once
    const int nw = wide_waves(hidden, kku);
    return 256 * (nw == 8 ? 1 : wide_wgs_per_cu(hidden / 64, kku)) / wide_streams_of(hidden, kku);
}
static int wide_grid(int64_t mb_rows, int hidden, int kku) {   // one workgroup per resident slot, fewer when the minibatch is small
    const int64_t st = wide_streams_of(hidden, kku), wgs = (mb_rows / 16 + st - 1) / st, slots = wide_slots(hidden, kku);
    return (int)(wgs < slots ? (wgs < 1 ? 1 : wgs) : slots);
}

}  // namespace pfa

using namespace pfa;

extern "C" size_t pfa_ppo_wide_workspace_bytes(const pfa_mlp_view *view) {
    int ktm, kku;
    bool col;
    if (wide_shape(view, &ktm, &col, &kku)) return 0;
    return align_up((size_t)wide_slots(view->hidden, kku) * wide_layout(view->hidden, ktm, col).kCount * sizeof(float), 256);
}

extern "C" int pfa_ppo_wide_supported(const pfa_mlp_view *view) {
    int ktm, kku;
    bool col;
    return wide_shape(view, &ktm, &col, &kku) == 0 ? 1 : 0;
}

extern "C" int pfa_ppo_wide_grad(const pfa_experience *exp, int64_t batch_rows, int32_t mb, const pfa_mlp_view *params, const pfa_mlp_view *grads,
                                 float *tail16, const pfa_ppo_hparams *hp, const double *adv_stats, int64_t global_mb_rows, void *workspace,
                                 pfa_stream_t stream) {
    int ktm, kku;
    bool col;
    if (int rc = wide_shape(params, &ktm, &col, &kku)) return rc;
    PFA_REQUIRE(exp && hp && grads && tail16 && workspace, "ppo.wide: null argument");
    PFA_REQUIRE(grads->w1 && grads->b1 && grads->w2 && grads->b2 && grads->wv && grads->bv && grads->ldw1 >= params->obs_dim,
                "ppo.wide: the gradient view must name the same six tensors");
    PFA_REQUIRE(hp->num_minibatches >= 1 && hp->bptt_horizon >= 1 && batch_rows % hp->num_minibatches == 0, "ppo.wide: bad minibatch partition");
    const int64_t mbs = batch_rows / hp->num_minibatches;
    PFA_REQUIRE(mbs % hp->bptt_horizon == 0 && mbs % 16 == 0, "ppo.wide: minibatch_size must be a multiple of bptt_horizon and of 16 (got %lld)", (long long)mbs);
    PFA_REQUIRE(mb >= 0 && mb < hp->num_minibatches && global_mb_rows >= mbs, "ppo.wide: minibatch index / global rows out of range");
    PFA_REQUIRE(exp->obs && exp->actions && exp->logprobs && exp->values && exp->advantages && exp->returns, "ppo.wide: null experience buffer");
    PFA_REQUIRE(!hp->norm_adv || adv_stats, "ppo.wide: norm_adv needs adv_stats");
    RowMap map{mb, hp->num_minibatches, hp->bptt_horizon};
    const MlpView pv{params->w1, params->ldw1, params->obs_dim, params->b1, params->w2, params->b2, params->wv, params->bv, params->num_actions, params->hidden};
    float *partials = (float *)workspace;
    const int grid = wide_grid(mbs, params->hidden, kku);
    {
        ScopedKernelTimer timer("ppo_wide_grad", (hipStream_t)stream);
#define PFA_WIDE_LAUNCH(DPV, KKUV, KTMV, COLV, HV)                                                                                        \
    {                                                                                                                                    \
        constexpr int NWV = wide_waves(HV, KKUV), MSV = HV / (16 * NWV);                                                                 \
        constexpr size_t lds_bytes = (size_t)WideLds<DPV, MSV, KKUV, NWV, KTMV>::kFloats * sizeof(float);                                \
        constexpr int STV = wide_streams(MSV, KKUV, NWV);                                      \
        static bool attr_set = false;                                                                                                    \
        if (!attr_set) {                                                                                                                 \
            PFA_CHECK_HIP(hipFuncSetAttribute((const void *)ppo_wide_grad_kernel<DPV, KKUV, KTMV, COLV, MSV, NWV>,                        \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));                              \
            attr_set = true;                                                                                                             \
        }                                                                                                                                \
        hipLaunchKernelGGL((ppo_wide_grad_kernel<DPV, KKUV, KTMV, COLV, MSV, NWV>), dim3(grid), dim3(64 * NWV * STV), lds_bytes,                \
                           (hipStream_t)stream, *exp, map, (long long)mbs, pv, *hp, adv_stats, (double)global_mb_rows, partials);        \
    }
#define PFA_WIDE_DP(HV)                                                                \
    if (col) PFA_WIDE_LAUNCH(64, 13, 3, true, HV)                                      \
    else if (params->obs_stride == 64) PFA_WIDE_LAUNCH(64, 16, 4, false, HV)           \
    else if (params->obs_stride == 32) PFA_WIDE_LAUNCH(32, 8, 2, false, HV)            \
    else PFA_WIDE_LAUNCH(16, 4, 1, false, HV)
        switch (params->hidden) {
            case 64: PFA_WIDE_DP(64) break;
            case 256: PFA_WIDE_DP(256) break;
            default: PFA_WIDE_DP(512) break;
        }
#undef PFA_WIDE_DP
#undef PFA_WIDE_LAUNCH
        PFA_LAUNCH_CHECK();
    }
    const WideLayout L = wide_layout(params->hidden, ktm, col);
    ScopedKernelTimer timer("ppo_wide_reduce", (hipStream_t)stream);
    hipLaunchKernelGGL(ppo_wide_reduce_kernel, dim3((L.kCount + 63) / 64), dim3(64 * kWideRedSl), 0, (hipStream_t)stream, partials, grid, L,
                       params->obs_dim, params->num_actions, params->hidden, (float *)grads->w1, grads->ldw1, (float *)grads->b1, (float *)grads->w2,
                       (float *)grads->b2, (float *)grads->wv, (float *)grads->bv, tail16);
    PFA_LAUNCH_CHECK();
    return 0;
}
