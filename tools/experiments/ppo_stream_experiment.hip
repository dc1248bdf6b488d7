// ppo_stream.hip — the minibatch step of clean_pufferl.train (clean_pufferl.py:175-246) for the MLP policy on observation rows
// of up to 64 floats, as ONE instruction stream per SIMD.
//
// Why a second form of the gradient kernel (measured on MI355X, tools/ubench/simd_share.cpp, profiles/r02_ubench.txt): an fp32
// MFMA and fp32 VALU work never overlap on a SIMD — not inside one wave and not between the two waves of a SIMD (MFMA | VALU =
// sum of the two, MFMA | MFMA = 2x) — and a wave that streams MFMAs back to back keeps its SIMD partner from issuing anything
// (its LDS-latency chain takes MFMA time + its own time, whatever the age or s_setprio of either wave).  The wave-pair kernel of
// ppo_update.hip therefore ran its producer and consumer one after the other, plus two workgroup barriers per tile (timeline:
// profiles/r02_grad_timeline_pairs.txt: consumer 7.9k cycles, then producer forward 6.7k, loss 2.6k with the matrix pipe idle,
// barriers 2.1k).  Here one wave per SIMD owns whole tiles and everything it needs lives in its 512 registers:
//
//   W1 (with b1 folded in as the weight of a constant-1 observation column), W2v as A fragments of the heads and as B
//   fragments of dh: 104 + 32 + 32 registers, built once per launch from an LDS copy of the parameters;
//   accumulators of dW1 (MFMA part 96, one trailing observation column on VALU 8), dW2v 32, db1 8, db2 4.
//
// Per 16-row tile: 13x8 forward + 32 heads + 32 dh + 96 dW1 + 32 dW2v = 296 MFMAs (the pair kernel: 328 at obs_dim 49), no
// workgroup barrier, LDS only for the three transpositions the weight gradients need (X tile, hidden tile, dout tile), the
// log-softmax reductions across lane groups on v_permlane16/32_swap instead of ds_bpermute.
//
// The optimizer step is folded into the NEXT launch's prologue: every workgroup recomputes clip_grad_norm_ + Adam for all ~9.5k
// parameters from the reduced gradient (bit-identical on every workgroup: same inputs, same order), keeps the new weights for
// itself and only workgroup 0 stores them (ping-pong parameter sets, so nobody reads what it writes).  Per optimizer step that
// leaves two launches: this kernel and the partial reduction.
#include <cmath>

#include "common.hpp"
#include "mlp_tile.hpp"
#include "lane_ops.hpp"
#include "ppo_tile.hpp"

namespace pfa {

constexpr int kStreamThreads = 1024;   // 16 waves: 4 tile streams x 4 hidden-unit slices, four waves per SIMD
// kTailFloats (ppo_tile.hpp): the loss sums at the end of the gradient bucket are (hi, lo) float pairs of f64 sums

// torch.optim.Adam, single-tensor path, the arithmetic of adam_clip_kernel (ppo_update.hip)
__device__ __forceinline__ float adam_update(float g, float clip, float &m, float &v, float w, const AdamConsts &k) {
    const float gi = g * clip;
    m = m + (1.0f - k.beta1) * (gi - m);              // exp_avg.lerp_(grad, 1 - beta1)
    v = v * k.beta2 + (1.0f - k.beta2) * gi * gi;     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(v) / k.bc2_sqrt + k.eps;
    return w + k.neg_step_size * m / denom;           // param.addcdiv_(exp_avg, denom, value=-step_size)
}

#ifdef PFA_PROBES
// tools/probe_grad.py stream-trace: s_memtime stamps of workgroup 0, [wave][slot]; slot 0..3 prologue/epilogue, 8 + 8*tile + k per tile
__device__ unsigned long long *g_trace_s = nullptr;
#define PFA_SSTAMP(slot)                                                                              \
    do {                                                                                              \
        if (g_trace_s && blockIdx.x == 0 && lane == 0 && (slot) < 128) g_trace_s[wv * 128 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PFA_SSTAMP(slot) do { } while (0)
#endif

// relu as ONE integer max on the bit pattern (negative floats, -0.0 included, are negative integers); fmaxf costs a
// canonicalisation on top (IEEE maxnum quiets signalling NaNs) and VALU time is MFMA time on this path
__device__ __forceinline__ float relu_bits(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

// LDS of one tile STREAM (4 per workgroup): what its four waves share
template <int DP>
struct StreamLds {
    static constexpr int XS = DP + 2;        // X tile row stride (floats): forward B-fragment reads conflict-free
    static constexpr int HS = kHidden + 4;   // hidden tile row stride
    static constexpr int DS = 20;            // row stride of the [row][o] tiles (out partials, dout)
    static constexpr int kX = 0;             // two X slots: tile j+1 is staged while tile j is still being read
    static constexpr int kH = 2 * 16 * XS;   // hidden[row][u] (post-relu): every wave writes its own 32 columns
    static constexpr int kD = kH + 16 * HS;  // dout[row][o], written by the loss wave
    static constexpr int kP = kD + 16 * DS;  // 4 partial out[row][o] tiles (the heads' K-split over the four waves)
    static constexpr int kStreamFloats = kP + 4 * 16 * DS;
};
// Behind the four stream areas: the forward A fragments of W1 (b1 folded in), lane-major so that four k-steps come back in
// one conflict-free ds_read_b128 — w1t[(m*64 + lane)*W1S + kk] = W1ext[u = 16m + c][k = 4kk + g].  (In registers they would
// be 26 of the 128 a wave has at four waves per SIMD.)
template <int KKU>
struct StreamW1 {
    static constexpr int W1S = (KKU + 3) / 4 * 4 + 4;   // lane stride: 16 B aligned, spreads a b128 read over all banks
    static constexpr int kFloats = kMT * 64 * W1S;
};

// One workgroup partial, native (fragment) order; ppo_reduce_stream_kernel undoes the permutation.
template <int KTM>
struct StreamLayout {
    static constexpr int kDw1 = 0;                          // ((kt*8+m)*4+r)*64 + lane -> dW1[u=16m+c][k=16kt+4g+r]
    static constexpr int kDw2 = KTM * kMT * 4 * 64;         // ((m*4+r)*64 + lane)      -> dW2v[o=c][u=16m+4g+r]
    static constexpr int kCol = kDw2 + kMT * 4 * 64;        // u                        -> dW1[u][k=16*KTM]
    static constexpr int kDb1 = kCol + kHidden;             // u
    static constexpr int kDb2 = kDb1 + kHidden;             // o (16)
    static constexpr int kMain = kDb2 + kOut;               // floats summed in f32
    static constexpr int kStats = kMain;                    // 8 doubles (16 floats)
    static constexpr int kCount = kMain + kTailFloats;
};

// Workgroup = 16 waves = 4 tile streams x 4 hidden-unit slices, four waves on every SIMD.  Wave (t, s) works on stream t's tile
// for hidden units 32s .. 32s+31 (m = 2s, 2s+1); per round (4 tiles):
//   forward own units (26 MFMAs), relu, keep its 32 hidden columns in LDS; heads K-slice (8 MFMAs) -> partial out[row][o] in LDS
//   one wave of the stream (rotating) sums the four partials, evaluates the loss (ppo_tile.hpp) and publishes dout[row][o]
//   dh own units (8), relu', db1, the trailing column; dW1 own units (8 KTM), dW2v own units (8)
// The four waves of a stream hand over through LDS counters, not workgroup barriers (see the loop), so streams run out of phase.
template <int DP, int KKU, int KTM, bool COL>
__global__ void __launch_bounds__(kStreamThreads) ppo_grad_stream_kernel(StreamArgs A) {
    using L = StreamLds<DP>;
    using SL = StreamLayout<KTM>;
    constexpr int XS = L::XS, HS = L::HS, DS = L::DS, V = DP / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = lane_id(), wv = wave_id(), c = lane & 15, g = lane >> 4;
    const int t = wv >> 2, sl = wv & 3;   // tile stream, hidden-unit slice (waves are dealt to SIMDs round-robin: a stream's four waves sit on four SIMDs)
    const int a = A.a;
    const MlpOffsets off = mlp_offsets(DP, a);
    const int count = off.count;
    PFA_SSTAMP(0);

    // ---------------------------------------------------------------------------------------------- prologue: parameters
    float *wl = lds;   // the parameter vector in LDS (flat layout), dead once the fragments are in registers
    {
        constexpr int NI = (kHidden * DP + kHidden + 16 * kHidden + 16 + kHidden + 1 + kStreamThreads - 1) / kStreamThreads;
        float wk[NI];
        if (A.pending) {   // optimizer step of the previous gradient: every workgroup computes it, workgroup 0 stores it
            __shared__ double sh_norm[kStreamThreads / 64];
            float gk[NI], mk[NI], vk[NI];
#pragma unroll
            for (int u = 0; u < NI; ++u) {   // every load is issued before the first use
                const int i = threadIdx.x + u * kStreamThreads;
                const bool ok = i < count;
                gk[u] = ok ? A.g[i] : 0.0f;
                mk[u] = ok ? A.m_in[i] : 0.0f;
                vk[u] = ok ? A.v_in[i] : 0.0f;
                wk[u] = ok ? A.w_in[i] : 0.0f;
            }
            double ss = 0.0;
#pragma unroll
            for (int u = 0; u < NI; ++u) ss += (double)gk[u] * (double)gk[u];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            if (lane == 0) sh_norm[wv] = ss;
            __syncthreads();
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < kStreamThreads / 64; ++w) tot += sh_norm[w];
            const float total_norm = (float)sqrt(tot);
            float clip = 1.0f;
            if (A.adam.max_grad_norm > 0.0f) {
                clip = A.adam.max_grad_norm / (total_norm + 1e-6f);   // clip_grad_norm_
                clip = clip > 1.0f ? 1.0f : clip;
            }
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                const int i = threadIdx.x + u * kStreamThreads;
                if (i < count) {
                    wk[u] = adam_update(gk[u], clip, mk[u], vk[u], wk[u], A.adam);
                    if (blockIdx.x == 0) {
                        A.w_out[i] = wk[u];
                        A.m_out[i] = mk[u];
                        A.v_out[i] = vk[u];
                    }
                }
            }
            if (blockIdx.x == 0 && threadIdx.x < 6 && A.losses_acc)
                A.losses_acc[threadIdx.x] += ((double)A.g[count + 2 * threadIdx.x] + (double)A.g[count + 2 * threadIdx.x + 1]) * A.loss_scale;
        } else {
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                const int i = threadIdx.x + u * kStreamThreads;
                wk[u] = i < count ? A.w_in[i] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = threadIdx.x + u * kStreamThreads;
            if (i < count) wl[i] = wk[u];
        }
    }
    __syncthreads();

    // operand fragments of this wave's 32 hidden units (mlp_tile.hpp conventions: A[i][k] lane supplies (i=c, k-slot g);
    // B[k][j] lane supplies (k-slot g, j=c))
    constexpr int W1S = StreamW1<KKU>::W1S;
    float *w1t = lds + 4 * L::kStreamFloats;   // forward A: W1[u=16m+c][k=4kk+g]; column obs_dim carries b1 (X has a constant 1 there)
    float w2a[2][4];    // heads   A: W2v[o=c][u=16m+4g+r]
    float w2b[2][4];    // dh      B: W2v[o=4g+r][u=16m+c]
#pragma unroll
    for (int mm = 0; mm < 2; ++mm) {
        const int m = 2 * sl + mm;
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // the slice's table: wave (t, sl) writes k-steps 4t .. 4t+3
            const int kk = 4 * t + q, k = 4 * kk + g, u = 16 * m + c;
            if (kk < W1S - 4) w1t[(m * 64 + lane) * W1S + kk] = kk >= KKU ? 0.0f : (k == A.obs_dim ? wl[off.b1 + u] : wl[off.w1 + u * DP + k]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int u1 = 16 * m + 4 * g + r, o2 = 4 * g + r, u2 = 16 * m + c;
            w2a[mm][r] = c < a ? wl[off.w2 + c * kHidden + u1] : (c == a ? wl[off.wv + u1] : 0.0f);
            w2b[mm][r] = o2 < a ? wl[off.w2 + o2 * kHidden + u2] : (o2 == a ? wl[off.wv + u2] : 0.0f);
        }
    }
    float bo[4];   // b2 | bv | 0 for o = 4g + r (the loss waves add them to the summed partials)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = 4 * g + r;
        bo[r] = o < a ? wl[off.b2 + o] : (o == a ? wl[off.bv] : 0.0f);
    }
    __syncthreads();   // wl is dead: the tile areas may overwrite it
    PFA_SSTAMP(1);

    // ---------------------------------------------------------------------------------------------- rounds of 4 tiles
    float *pl = lds + t * L::kStreamFloats;
    float *hs = pl + L::kH, *ds = pl + L::kD, *po = pl + L::kP;
    const long long tiles = A.mb_rows / 16;
    const long long stream_global = (long long)blockIdx.x * 4 + t, stream_count = (long long)gridDim.x * 4;
    const int J = stream_global < tiles ? (int)((tiles - stream_global + stream_count - 1) / stream_count) : 0;   // per stream
    const bool aligned = (A.map.horizon & 15) == 0;
    const pfa_experience &ex = A.ex;

    float adv_mean = 0.0f, adv_den = 1.0f;
    if (A.hp.norm_adv) {   // clean_pufferl.py:211-213: unbiased std over the GLOBAL minibatch
        const double s1 = A.adv_stats[2 * A.map.mb], s2 = A.adv_stats[2 * A.map.mb + 1];
        const double mean = s1 / A.global_rows;
        double var = (s2 - s1 * mean) / (A.global_rows - 1.0);
        var = var > 0.0 ? var : 0.0;
        adv_mean = (float)mean;
        adv_den = (float)sqrt(var) + 1e-8f;
    }
    const float inv_rows = (float)(1.0 / A.global_rows);

    f32x4 acc1[KTM][2], acc2[2];
    float acc_col[2], db1[2], db2 = 0.0f, stats[6];
#pragma unroll
    for (int mm = 0; mm < 2; ++mm) {
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) acc1[kt][mm] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc2[mm] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc_col[mm] = db1[mm] = 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) stats[i] = 0.0f;

    // A stream's four waves synchronise among themselves through three LDS counters (no workgroup barrier in the loop, so the
    // four streams drift apart and one stream's loss runs under the others' MFMAs):
    //   ctr[0] += 1 by every wave once its out partial of the round is in LDS        -> the loss wave waits for 4 (j + 1)
    //   ctr[1]  = j + 1 by the loss wave once dout of round j is in LDS              -> everybody waits for it before the backward
    //   ctr[2] += 1 by every wave once its share of X(j + 1) is staged                -> the forward of round j + 1 waits for 4 (j + 2)
    // Reuse hazards: X slot (j+1)&1 is written only after ctr[1] >= j+1, i.e. after every wave finished forward(j) and with it
    // backward(j-1), the last reader of that slot; the partial tiles of round j+1 are written after backward(j), the dout tile of
    // round j+1 after all four partials of round j+1, i.e. after every backward(j).
    unsigned *ctr = reinterpret_cast<unsigned *>(w1t + StreamW1<KKU>::kFloats) + 4 * t;
    auto signal_add = [&](unsigned *p) {
        if (lane == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto wait_ge = [&](unsigned *p, unsigned target) {
        for (int spin = 0; spin < (1 << 22); ++spin) {   // bounded: a protocol bug must not hang the GPU
            // one LDS word, the same for every lane: decide on the scalar unit
            if ((unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) >= target) break;
            __builtin_amdgcn_s_sleep(1);
        }
    };

    // X staging: the stream's four waves load the 16 x V float4 of a tile, one (or none) per lane
    const int xidx = sl * 64 + lane;
    const bool xload = xidx < 16 * V;
    const int xr = xidx / V, xc4 = xidx - xr * V;
    const int bias_c4 = A.obs_dim >> 2, bias_comp = A.obs_dim & 3;
    float4 xpre = make_float4(0.f, 0.f, 0.f, 0.f);
    auto prefetch = [&](long long tile) {
        if (xload && tile < tiles) {
            const unsigned first = A.map.tile_first((unsigned)tile);
            const unsigned row = A.map.tile_row((unsigned)tile, first, xr, aligned);
            xpre = *reinterpret_cast<const float4 *>(ex.obs + (size_t)row * DP + 4 * xc4);
        }
    };
    auto stage = [&](float *xs) {   // registers -> LDS X slot; the constant-1 column that carries b1 goes in on the way
        if (xload) {
            float4 x = xpre;
            if (xc4 == bias_c4) {
                x.x = bias_comp == 0 ? 1.0f : x.x;
                x.y = bias_comp == 1 ? 1.0f : x.y;
                x.z = bias_comp == 2 ? 1.0f : x.z;
                x.w = bias_comp == 3 ? 1.0f : x.w;
            }
            float2 *d = reinterpret_cast<float2 *>(xs + xr * XS + 4 * xc4);
            d[0] = make_float2(x.x, x.y);
            d[1] = make_float2(x.z, x.w);
        }
    };
    prefetch(stream_global);
    stage(pl + L::kX);
    if (threadIdx.x < 16) reinterpret_cast<unsigned *>(w1t + StreamW1<KKU>::kFloats)[threadIdx.x] = (threadIdx.x & 3) == 2 ? 4u : 0u;   // X(0) staged
    __syncthreads();

    for (int j = 0; j < J; ++j) {
        const float *xs = pl + L::kX + (j & 1) * 16 * XS;
        const long long tile = stream_global + (long long)j * stream_count;
        const bool loss_wave = sl == ((j + t) & 3);   // rotates over the stream's waves, i.e. over the SIMDs
        PFA_SSTAMP(8 + 8 * j);
        prefetch(tile + stream_count);   // next round's X: in flight until it is staged behind the loss
        RowScalars rs{0, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (loss_wave) {   // per-row scalars (lane c <-> row c, replicated over the 4 lane groups)
            const unsigned first = A.map.tile_first((unsigned)tile);
            const unsigned fr = A.map.tile_row((unsigned)tile, first, c, aligned);
            rs = RowScalars{ex.actions[fr], ex.logprobs[fr], ex.values[fr], ex.advantages[fr], ex.returns[fr], 1.0f};
        }
        wait_ge(ctr + 2, 4u * (unsigned)(j + 1));   // X(j) staged by all four waves
        // ---- forward, own units: hidden^T[u][row] = W1ext . Xext^T --------------------------------------------------------
        f32x4 h[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int k4 = 0; k4 < (KKU + 3) / 4; ++k4) {
            const f32x4 wq0 = *reinterpret_cast<const f32x4 *>(w1t + ((2 * sl) * 64 + lane) * W1S + 4 * k4);
            const f32x4 wq1 = *reinterpret_cast<const f32x4 *>(w1t + ((2 * sl + 1) * 64 + lane) * W1S + 4 * k4);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * k4 + q < KKU) {
                    const float xb = xs[c * XS + 4 * (4 * k4 + q) + g];
                    h[0] = mfma16(wq0[q], xb, h[0]);
                    h[1] = mfma16(wq1[q], xb, h[1]);
                }
        }
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
#pragma unroll
            for (int r = 0; r < 4; ++r) h[mm][r] = relu_bits(h[mm][r]);
            *reinterpret_cast<f32x4 *>(hs + c * HS + 16 * (2 * sl + mm) + 4 * g) = h[mm];   // hidden[row = c][u]: own columns
        }
        // ---- heads, K-slice over own units: partial out^T[o][row], stored as out[row = c][o = 4g + r] ---------------------
        {
            f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o0 = mfma16(w2a[0][r], h[0][r], o0);
                o1 = mfma16(w2a[1][r], h[1][r], o1);
            }
            *reinterpret_cast<f32x4 *>(po + sl * 16 * DS + c * DS + 4 * g) = o0 + o1;
        }
        signal_add(ctr + 0);
        PFA_SSTAMP(8 + 8 * j + 1);
        if (loss_wave) {
            wait_ge(ctr + 0, 4u * (unsigned)(j + 1));
            PFA_SSTAMP(8 + 8 * j + 2);
            const f32x4 p0 = *reinterpret_cast<const f32x4 *>(po + 0 * 16 * DS + c * DS + 4 * g);
            const f32x4 p1 = *reinterpret_cast<const f32x4 *>(po + 1 * 16 * DS + c * DS + 4 * g);
            const f32x4 p2 = *reinterpret_cast<const f32x4 *>(po + 2 * 16 * DS + c * DS + 4 * g);
            const f32x4 p3 = *reinterpret_cast<const f32x4 *>(po + 3 * 16 * DS + c * DS + 4 * g);
            const f32x4 out = ((p0 + p1) + (p2 + p3)) + f32x4{bo[0], bo[1], bo[2], bo[3]};   // out^T[o = 4g + r][row = c]
            const LossOut lo = ppo_loss_tile<false, true>(out, rs, a, 0u, g, A.hp, adv_mean, adv_den, inv_rows);
            *reinterpret_cast<f32x4 *>(ds + c * DS + 4 * g) = lo.dout;   // dout[row = c][o]
            stats[0] += lo.pg;
            stats[1] += lo.v_loss;
            stats[2] += lo.ent;
            stats[3] += lo.neg_logratio;
            stats[4] += lo.kl;
            stats[5] += lo.clipped;
            if (lane == 0) __hip_atomic_store(ctr + 1, (unsigned)(j + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        wait_ge(ctr + 1, (unsigned)(j + 1));   // dout(j) published
        PFA_SSTAMP(8 + 8 * j + 3);
        stage(pl + L::kX + ((j + 1) & 1) * 16 * XS);
        signal_add(ctr + 2);
        PFA_SSTAMP(8 + 8 * j + 4);
        // ---- backward, own units.  hidden back as [row = 4g+r][u = 16m+c]: relu' mask in dh's layout AND the A of dW2v ----
        {
            const f32x4 dout = *reinterpret_cast<const f32x4 *>(ds + c * DS + 4 * g);   // dout^T[o = 4g+r][row = c]: A of dh (A = C^T)
            float xa[KTM][4], xc[4], dfrag[4], hrow[2][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int kt = 0; kt < KTM; ++kt) xa[kt][r] = xs[(4 * g + r) * XS + 16 * kt + c];   // X^T: A[i = k][k-slot = row]
                xc[r] = COL ? xs[(4 * g + r) * XS + 16 * KTM] : 0.0f;
                dfrag[r] = ds[(4 * g + r) * DS + c];                                            // dout[row = 4g+r][o = c]: B of dW2v
                hrow[0][r] = hs[(4 * g + r) * HS + 16 * (2 * sl) + c];
                hrow[1][r] = hs[(4 * g + r) * HS + 16 * (2 * sl + 1) + c];
            }
            f32x4 dh[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int r = 0; r < 4; ++r) {   // dh[row][u] = dout . W2v
                dh[0] = mfma16(dout[r], w2b[0][r], dh[0]);
                dh[1] = mfma16(dout[r], w2b[1][r], dh[1]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)   // dW2v^T[u][o] += hidden^T . dout (independent of dh: fills its latency)
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) acc2[mm] = mfma16(hrow[mm][r], dfrag[r], acc2[mm]);
            if (sl == 0) db2 += (dfrag[0] + dfrag[1]) + (dfrag[2] + dfrag[3]);   // column sums of dout: lane (c = o) over its rows
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dh[mm][r] = hrow[mm][r] > 0.0f ? dh[mm][r] : 0.0f;   // relu'
                    db1[mm] += dh[mm][r];
                    if (COL) acc_col[mm] = fmaf(xc[r], dh[mm][r], acc_col[mm]);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r)   // dW1^T[k][u] += X^T . dh (B = dh's own C fragment)
#pragma unroll
                for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) acc1[kt][mm] = mfma16(xa[kt][r], dh[mm][r], acc1[kt][mm]);
        }
    }

    // ---------------------------------------------------------------------------------------------- epilogue: one partial
    PFA_SSTAMP(2);
    __syncthreads();   // every wave is done with the tile areas
    {
        float *img = lds + (size_t)t * SL::kMain;   // one image per stream, then 1024 threads sum the four
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
            const int m = 2 * sl + mm;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int kt = 0; kt < KTM; ++kt) img[SL::kDw1 + ((kt * kMT + m) * 4 + r) * 64 + lane] = acc1[kt][mm][r];
                img[SL::kDw2 + (m * 4 + r) * 64 + lane] = acc2[mm][r];
            }
            const float sc = gsum<true>(acc_col[mm]), sb = gsum<true>(db1[mm]);   // per-lane partials over rows 4g+r: sum over g
            if (g == 0) {
                img[SL::kCol + 16 * m + c] = sc;
                img[SL::kDb1 + 16 * m + c] = sb;
            }
        }
        if (sl == 0) {
            const float s2 = gsum<true>(db2);
            if (g == 0) img[SL::kDb2 + c] = s2;
        }
        double *st = reinterpret_cast<double *>(lds + 4 * (size_t)SL::kMain);   // [wave][8] behind the images
#pragma unroll
        for (int i = 0; i < 6; ++i) {   // loss sums: per-lane f32 over the wave's few loss rounds, everything above that in f64
            double sv = (double)stats[i];   // every lane group holds a copy of the per-row terms: lanes 0..15
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sv += __shfl_xor(sv, o, 64);
            if (lane == 0) st[wv * kNumStats + i] = sv;
        }
        if (lane == 0) st[wv * kNumStats + 6] = st[wv * kNumStats + 7] = 0.0;
    }
    __syncthreads();
    float *dst = A.partials + (size_t)blockIdx.x * SL::kCount;
    for (int i = threadIdx.x; i < SL::kMain; i += kStreamThreads)
        dst[i] = (lds[i] + lds[SL::kMain + i]) + (lds[2 * SL::kMain + i] + lds[3 * SL::kMain + i]);
    if (threadIdx.x < kNumStats) {
        const double *st = reinterpret_cast<const double *>(lds + 4 * (size_t)SL::kMain);
        double sv = 0.0;
        for (int w = 0; w < kStreamThreads / 64; ++w) sv += st[w * kNumStats + threadIdx.x];
        reinterpret_cast<double *>(dst + SL::kStats)[threadIdx.x] = sv;
    }
    PFA_SSTAMP(3);
}

// Fixed-order sum of the workgroup partials + scatter into the flat gradient layout; the loss sums are carried in f64 and leave
// as (hi, lo) float pairs behind the gradient (the bucket a data-parallel all-reduce sums in f32: hi + lo keeps ~48 bits).
// Block = 32 native slots x 8 slices of the partial index; the last block owns the loss sums.
template <int DP, int KTM, bool COL>
__global__ void __launch_bounds__(256) ppo_reduce_stream_kernel(const float *partials, int nparts, int a, int obs_dim, float *grads) {
    using SL = StreamLayout<KTM>;
    const MlpOffsets off = mlp_offsets(DP, a);
    constexpr int kBlocksMain = (SL::kMain + 31) / 32;
    if ((int)blockIdx.x == kBlocksMain) {   // loss sums, f64: thread = (partial slice, statistic), then a fixed-order tree
        __shared__ double shs[32][kNumStats];
        const int st = threadIdx.x & (kNumStats - 1), slice = threadIdx.x >> 3;
        double s = 0.0;
        for (int i = slice; i < nparts; i += 32) s += reinterpret_cast<const double *>(partials + (size_t)i * SL::kCount + SL::kStats)[st];
        shs[slice][st] = s;
        __syncthreads();
        if (threadIdx.x < kNumStats) {
            double t = 0.0;
            for (int q = 0; q < 32; ++q) t += shs[q][threadIdx.x];
            const float hi = (float)t;
            grads[off.count + 2 * threadIdx.x] = hi;
            grads[off.count + 2 * threadIdx.x + 1] = (float)(t - (double)hi);
        }
        return;
    }
    __shared__ float sh[8][32];
    const int ql = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int q = blockIdx.x * 32 + ql;
    float acc = 0.0f;
    if (q < SL::kMain) {
        for (int i0 = sl; i0 < nparts; i0 += 128) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = i0 + 8 * u;
                v[u] = i < nparts ? partials[(size_t)i * SL::kCount + q] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += v[u];
        }
    }
    sh[sl][ql] = acc;
    __syncthreads();
    if (sl != 0 || q >= SL::kMain) return;
    const float s = ((sh[0][ql] + sh[1][ql]) + (sh[2][ql] + sh[3][ql])) + ((sh[4][ql] + sh[5][ql]) + (sh[6][ql] + sh[7][ql]));
    int p = -1;
    if (q < SL::kDw2) {
        const int ln = q & 63, r = (q >> 6) & 3, m = (q >> 8) & (kMT - 1), kt = q >> 11;
        const int k = 16 * kt + 4 * (ln >> 4) + r;
        if (k < obs_dim) p = off.w1 + (16 * m + (ln & 15)) * DP + k;   // k == obs_dim is the constant-1 column (= db1), beyond: padding
    } else if (q < SL::kCol) {
        const int t = q - SL::kDw2, ln = t & 63, r = (t >> 6) & 3, m = t >> 8;
        const int o = ln & 15, u = 16 * m + 4 * (ln >> 4) + r;
        if (o < a) p = off.w2 + o * kHidden + u;
        else if (o == a) p = off.wv + u;
    } else if (q < SL::kDb1) {
        if (COL && 16 * KTM < obs_dim) p = off.w1 + (q - SL::kCol) * DP + 16 * KTM;
    } else if (q < SL::kDb2) {
        p = off.b1 + (q - SL::kDb1);
    } else {
        const int o = q - SL::kDb2;
        if (o < a) p = off.b2 + o;
        else if (o == a) p = off.bv;
    }
    if (p >= 0) grads[p] = s;
}

// The LAST optimizer step of an update (no following gradient launch to fold it into): clip + Adam from the reduced gradient,
// result into the trainer's own parameter / moment buffers (element-wise, so in == out is fine when the call has one step).
constexpr int kAdamFinishThreads = 256;
__global__ void __launch_bounds__(kAdamFinishThreads) adam_finish_kernel(const float *w_in, const float *m_in, const float *v_in, const float *g,
                                                                     float *w_out, float *m_out, float *v_out, int count, AdamConsts k,
                                                                     double *losses_acc, double loss_scale) {
    // every workgroup derives the same norm (same values, same order as the prologue above) and updates its own 256-slot slice
    __shared__ double sh_norm[kAdamFinishThreads / 64];
    double ss = 0.0;
    for (int i0 = threadIdx.x; i0 < count; i0 += 8 * kAdamFinishThreads) {
        float gv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * kAdamFinishThreads;
            gv[u] = i < count ? g[i] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) ss += (double)gv[u] * (double)gv[u];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (lane_id() == 0) sh_norm[wave_id()] = ss;
    __syncthreads();
    const double tot = (sh_norm[0] + sh_norm[1]) + (sh_norm[2] + sh_norm[3]);
    const float total_norm = (float)sqrt(tot);
    float clip = 1.0f;
    if (k.max_grad_norm > 0.0f) {
        clip = k.max_grad_norm / (total_norm + 1e-6f);
        clip = clip > 1.0f ? 1.0f : clip;
    }
    const int i = blockIdx.x * kAdamFinishThreads + threadIdx.x;
    if (i < count) {
        float m = m_in[i], v = v_in[i];
        w_out[i] = adam_update(g[i], clip, m, v, w_in[i], k);
        m_out[i] = m;
        v_out[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < 6 && losses_acc)
        losses_acc[threadIdx.x] += ((double)g[count + 2 * threadIdx.x] + (double)g[count + 2 * threadIdx.x + 1]) * loss_scale;
}

struct StreamShape {   // which instantiation serves (obs_stride, obs_dim); kind 0 = none (the wave-pair kernel takes it)
    int kind;
};
static int stream_kind(const pfa_mlp_dims *d) {
    if (d->heads != 0 || d->hidden != kHidden) return 0;
    if (d->obs_stride == 64 && d->obs_dim == 49) return 1;   // 7x7 grid: 12 k-steps of data + the bias column, 3 k-tiles + column 48
    if (d->obs_stride == 32 && d->obs_dim == 25) return 2;   // 5x5
    if (d->obs_stride == 16 && d->obs_dim == 9) return 3;    // 3x3
    return 0;
}
static int stream_grid(int64_t mb_rows) {
    const int64_t tiles = mb_rows / 16, wgs = (tiles + 3) / 4;
    return (int)(wgs < 256 ? (wgs < 1 ? 1 : wgs) : 256);
}
static size_t stream_partial_floats(int kind) {
    return kind == 1 ? StreamLayout<3>::kCount : kind == 2 ? StreamLayout<2>::kCount : StreamLayout<1>::kCount;
}

static AdamConsts adam_consts(float lr, float beta1, float beta2, float eps, int64_t step, float max_grad_norm) {
    // torch.optim.Adam (single tensor): bias corrections and step size are python floats (f64)
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
    return AdamConsts{(float)(-(double)lr / bc1), (float)std::sqrt(bc2), beta1, beta2, eps, max_grad_norm};
}

template <int DP, int KKU, int KTM, bool COL>
static int launch_stream(const StreamArgs &args, int grid, hipStream_t stream) {
    using SL = StreamLayout<KTM>;
    constexpr size_t lds_loop = ((size_t)4 * StreamLds<DP>::kStreamFloats + StreamW1<KKU>::kFloats + 16) * sizeof(float);
    constexpr size_t lds_epi = (size_t)4 * SL::kMain * sizeof(float) + (kStreamThreads / 64) * kNumStats * sizeof(double);
    constexpr size_t lds_par = (size_t)(kHidden * DP + kHidden + 16 * kHidden + 16 + kHidden + 1) * sizeof(float);
    constexpr size_t lds_bytes = lds_loop > lds_epi ? (lds_loop > lds_par ? lds_loop : lds_par) : (lds_epi > lds_par ? lds_epi : lds_par);
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    if (!attr_set) {
        PFA_CHECK_HIP(hipFuncSetAttribute((const void *)ppo_grad_stream_kernel<DP, KKU, KTM, COL>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_set = true;
    }
    hipLaunchKernelGGL((ppo_grad_stream_kernel<DP, KKU, KTM, COL>), dim3(grid), dim3(kStreamThreads), lds_bytes, stream, args);
    PFA_LAUNCH_CHECK();
    return 0;
}

// gradient launch + partial reduction for one minibatch; `args` carries the parameter source (plain or pending Adam)
int stream_grad_step(int kind, StreamArgs args, int64_t mbs, const pfa_mlp_dims *dims, float *grads, hipStream_t stream) {
    const int grid = stream_grid(mbs);
    {
        ScopedKernelTimer timer("ppo_mlp_grad", stream);
        const int rc = kind == 1   ? launch_stream<64, 13, 3, true>(args, grid, stream)
                       : kind == 2 ? launch_stream<32, 7, 2, false>(args, grid, stream)
                                   : launch_stream<16, 3, 1, false>(args, grid, stream);
        if (rc) return rc;
    }
    ScopedKernelTimer timer2("ppo_reduce", stream);
    const int a = dims->num_actions, od = dims->obs_dim;
    if (kind == 1)
        hipLaunchKernelGGL((ppo_reduce_stream_kernel<64, 3, true>), dim3((StreamLayout<3>::kMain + 31) / 32 + 1), dim3(256), 0, stream,
                           args.partials, grid, a, od, grads);
    else if (kind == 2)
        hipLaunchKernelGGL((ppo_reduce_stream_kernel<32, 2, false>), dim3((StreamLayout<2>::kMain + 31) / 32 + 1), dim3(256), 0, stream,
                           args.partials, grid, a, od, grads);
    else
        hipLaunchKernelGGL((ppo_reduce_stream_kernel<16, 1, false>), dim3((StreamLayout<1>::kMain + 31) / 32 + 1), dim3(256), 0, stream,
                           args.partials, grid, a, od, grads);
    PFA_LAUNCH_CHECK();
    return 0;
}

#ifdef PFA_PROBES
int stream_set_trace(unsigned long long *buf) {
    PFA_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_s), &buf, sizeof(buf)));
    return 0;
}
#endif

int stream_kind_of(const pfa_mlp_dims *dims) { return stream_kind(dims); }
size_t stream_partials_bytes(const pfa_mlp_dims *dims) { return align_up((size_t)256 * stream_partial_floats(stream_kind(dims)) * sizeof(float), 256); }

int stream_adam_finish(const float *w_in, const float *m_in, const float *v_in, const float *g, float *w_out, float *m_out, float *v_out,
                       int count, float lr, float beta1, float beta2, float eps, int64_t step, float max_grad_norm, double *losses_acc,
                       double loss_scale, hipStream_t stream) {
    ScopedKernelTimer timer("adam_clip", stream);
    hipLaunchKernelGGL(adam_finish_kernel, dim3((count + kAdamFinishThreads - 1) / kAdamFinishThreads), dim3(kAdamFinishThreads), 0, stream, w_in, m_in, v_in, g, w_out, m_out, v_out, count,
                       adam_consts(lr, beta1, beta2, eps, step, max_grad_norm), losses_acc, loss_scale);
    PFA_LAUNCH_CHECK();
    return 0;
}

AdamConsts stream_adam_consts(float lr, float beta1, float beta2, float eps, int64_t step, float max_grad_norm) {
    return adam_consts(lr, beta1, beta2, eps, step, max_grad_norm);
}

}  // namespace pfa
