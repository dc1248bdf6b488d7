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
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include <hip/hip_ext.h>

#include <unordered_map>

#include "common.hpp"
#include "mlp_tile.hpp"
#include "sampler.hpp"
#include "ppo_tile.hpp"
#include "p2p_ll.hpp"

#ifndef PFA_GRAD_FCOL
#define PFA_GRAD_FCOL 1   // 0 = the trailing column through a full MFMA k-step in the forward (A/B timing)
#endif
#ifndef PFA_GRAD_GLDS
// 1 = the next X tile by direct-to-LDS loads (global_load_lds_dwordx4, swizzled chunk layout, third tile slot) instead of the
// 16-register prefetch + 8 ds_writes.  Built and measured in round 4 (profiles/r04_grad_variants.txt): identical bits, 54.75 vs
// 54.73 us per launch — the register allocation is set by the consumer branch (dW1 accumulators), so the producer's freed
// prefetch registers change neither the occupancy nor the three loop-invariant dwords that spill.  Kept as a variant, off.
#define PFA_GRAD_GLDS 0
#endif
#ifndef PFA_GRAD_PIPE
#define PFA_GRAD_PIPE 1   // 0 = the round-2 instruction order (kept for A/B timing with tools/variant_bench.py)
#endif

namespace pfa {

// (producer, consumer) wavefront pairs per workgroup.  Rows of up to 64 floats: 4 pairs = 8 waves, two per SIMD, two workgroups
// per CU.  Rows of 96 / 128 floats: the W1 fragment table (57 / 74 KB) and the consumer's dW1 accumulators (192 / 256 registers)
// leave room for 2 pairs = 4 waves, one per SIMD with the whole register file, one workgroup per CU.
constexpr int grad_pairs(int dp) { return dp <= 64 ? 4 : 2; }
constexpr int grad_threads(int dp) { return grad_pairs(dp) * 2 * 64; }
constexpr int grad_waves_per_simd(int dp) { return dp <= 64 ? 2 : 1; }

// LDS map of kernel A (floats).  A workgroup is 4 PAIRS of wavefronts; wave p (producer) and wave p+4 (consumer) share a
// SIMD, so one wave's LDS waits / VALU stretches (the loss math) are covered by the other's MFMAs (two waves per SIMD fit
// because neither role needs more than ~230 registers: the producer keeps no dW1 accumulators, the consumer no forward
// state).  Per pair: two X tiles (ring), the hidden tile and the dout tile of the tile in flight between the two waves.
// Shared by all waves: lane-major MFMA fragment tables, every fragment read one conflict-free ds_read_b128:
//   w1t[m][lane][KS(+4)] A frags of the forward GEMM: W1[16m + c][4kk + g]
//   b1t[m][lane][4] accumulator init   w2t[m][lane][4] A frags of the heads   w2bt[m][lane][4] B frags of dh = dout.W2v
// GL (64-float rows in the pipelined form): the X tiles are filled by direct-to-LDS loads (global_load_lds_dwordx4: the data never
// passes through registers).  Such a load writes wave-uniform base + lane x 16 B, so the tile is stored as 16-byte chunks in the
// order the lanes issue them, and which (row, column quad) a lane fetches is chosen so that both fragment reads stay free of bank
// conflicts: chunk (row, kk) — columns 4 kk .. 4 kk + 3 of a row — sits in 1 KB block kk / 4 at position
//   16 (kk & 3) + (row & 12) + ((row + kk) & 3)
// (the forward's B fragment reads 16 rows x one kk per instruction, the consumer's A fragment 4 rows x 4 kk: either way the 64
// lanes hit 64 different banks; the padded [row][DP + 2] tile of the register-staged form has 2-way conflicts on the second).
// A third slot: the load of X(j+1) is issued at the top of tile j, while the consumer still reads X(j-1).
constexpr bool grad_glds(int dp, int ktm) { return PFA_GRAD_GLDS && PFA_GRAD_PIPE && dp == 64 && ktm <= 3; }
template <int DP, bool GL = false>
struct GradLds {
    static constexpr int XS = XTile<DP>::XS;
    static constexpr int KS = DP / 4;
    static constexpr int W1S = KS + 4;        // lane stride of w1t (16 B aligned, spreads 16 lanes over all banks)
    static constexpr int HS = kHidden + 4;    // row stride of the hidden tile (16 B aligned rows, conflict-free reads)
    static constexpr int DS = 20;             // dout tile row stride
    static constexpr int kXSlot = GL ? 16 * DP : 16 * XS;   // floats per X tile
    static constexpr int kXSlots = GL ? 3 : 2;
    static constexpr int kXT = 0;                        // + slot * kXSlot          X tiles (ring)
    static constexpr int kHT = kXSlots * kXSlot;         // hidden tile [row][u] (post-relu) of the published tile
    static constexpr int kDT = kHT + 16 * HS;            // dout tile [row][o] of the published tile
    static constexpr int kPairFloats = kDT + 16 * DS;
    static constexpr int kW1Floats = kMT * 64 * W1S;
    static constexpr int kTabFloats = kMT * 64 * 4;
    static constexpr int kFloats = grad_pairs(DP) * kPairFloats + kW1Floats + 3 * kTabFloats;
    // float index of X[row][col] inside a tile
    __device__ static __forceinline__ int xi(int row, int col) {
        if constexpr (GL) {
            const int kk = col >> 2;
            return (kk >> 2) * 256 + (((kk & 3) << 4) + (row & 12) + ((row + kk) & 3)) * 4 + (col & 3);
        } else {
            return row * XS + col;
        }
    }
};

// "Native" layout of one workgroup partial: gradients in MFMA C-fragment order (conflict-free LDS reduction and
// coalesced global traffic); ppo_reduce_kernel undoes the permutation once.
template <int DP, int KTM = DP / 16, bool COL = false>
struct NativeLayout {
    // only what the instantiation computes travels: KTM k-tiles of dW1 and, with COL, the one trailing column (the 7x7 grid's
    // column 48) — the remaining k-tiles are observation padding, their gradient is zero and nobody writes or sums them
    static constexpr int kDw1 = 0;                           // ((kt*8+m)*4+r)*64 + lane -> W1[16m+c][16kt+4g+r], kt < KTM
    static constexpr int kCol = kDw1 + KTM * kMT * 4 * 64;   // u = 16m + c              -> W1[u][16 KTM]         (COL)
    static constexpr int kDw2 = kCol + (COL ? kHidden : 0);  // ((m*4+r)*64 + lane)      -> W2v[o=c][16m+4g+r]
    static constexpr int kDb1 = kDw2 + kMT * 4 * 64;         // u
    static constexpr int kDb2 = kDb1 + kHidden;              // o (16)
    static constexpr int kStats = kDb2 + kOut;               // 8
    static constexpr int kCount = kStats + kNumStats;
};

}  // namespace pfa
#include "ppo_bf16.hpp"   // the same step on the bf16 matrix path (opt-in product form); needs NativeLayout
namespace pfa {

#ifdef PFA_PROBES
// tools/probe_grad.py --trace: s_memtime stamps of workgroup 0 (lane 0 of every wave), [wave][tile][8]
__device__ unsigned long long *g_trace = nullptr;
__device__ int g_trace_tiles = 0;
#define PFA_STAMP(j, k)                                                                                         \
    do {                                                                                                        \
        if (g_trace && blockIdx.x == 0 && lane == 0 && (j) < g_trace_tiles)                                     \
            g_trace[((size_t)wv * g_trace_tiles + (j)) * 8 + (k)] = __builtin_amdgcn_s_memtime();               \
    } while (0)
#else
#define PFA_STAMP(j, k) do { } while (0)
#endif

// ABL: ablation mask for tools/probe_grad.py (-DPFA_PROBES builds only; the product always runs ABL = 0):
//   1 skip the loss math   2 skip the consumer's dW1 MFMAs   4 skip the forward MFMAs
//
// Schedule of one pair over its tiles j = 0..J-1 (two workgroup barriers per tile, alpha_j and beta_j):
//   producer   stage X(j), forward, heads, loss        | alpha_j | hidden(j), dout(j) -> LDS | beta_j | dW2v(j) from LDS ...
//   consumer   dh(j-1) = dout.W2v, relu', db1, dW1(j-1)  | alpha_j |                           | beta_j | dh(j), ... 
// i.e. the consumer's whole share of tile j (160 MFMAs) runs under the producer's forward + loss of tile j+1 (192 MFMAs + the
// VALU-heavy loss).  alpha_j: the consumer is done reading hidden/dout(j-1), so they may be overwritten; beta_j: published.
// dh's C fragment (rows 4g+r, column u = c) IS the B fragment dW1 = X^T dh needs, so the consumer never stages dh.
// KTM / COL: dW1 = X^T dh is contracted on MFMA for the first KTM 16-column tiles of X only; COL adds ONE trailing column
// (k = 16 KTM) as 32 VALU fmas per tile.  Columns beyond that are observation padding whose gradient is 0 by construction.
// The 7x7 grid (49 = 3 x 16 + 1 columns) runs <KTM = 3, COL = true>: 32 MFMAs (1024 matrix-pipe cycles) less per tile.
// PERM: the head outputs in permuted fragment rows (ppo_tile.hpp: ppo_loss_tile<.., PERM>): dh takes three k-steps, not four.
__host__ __device__ constexpr int slot_output(bool perm, int slot) {   // fragment row `slot` -> logical output (>= 16: padding)
    return !perm ? slot : ((slot & 3) == 3 ? 99 : 3 * (slot >> 2) + (slot & 3));
}
template <int DP, int ABL = 0, int KKU = DP / 4, bool MH = false, int KTM = DP / 16, bool COL = false, bool PERM = false>
__global__ void __launch_bounds__(grad_threads(DP), grad_waves_per_simd(DP))
    ppo_mlp_grad_kernel(pfa_experience ex, RowMap map, long long mb_rows, const float *params, int a, uint32_t heads,
                        pfa_ppo_hparams hp, const double *adv_stats /* [nmb][2] */, double global_rows, float *partials) {
    constexpr bool kGlds = grad_glds(DP, KTM);
    using L = GradLds<DP, kGlds>;
    using NL = NativeLayout<DP, KTM, COL>;
    constexpr int XS = L::XS, HS = L::HS, DS = L::DS, KS = DP / 4, V = DP / 4, W1S = L::W1S;
    constexpr int NLD = (16 * V + 63) / 64;  // float4 loads per lane per tile
    constexpr int kGradPairs = grad_pairs(DP), kGradThreads = grad_threads(DP);
    constexpr bool kQuad = kGradPairs == 4 && 4 * NL::kCount <= L::kFloats;   // epilogue: four reduction buffers side by side
    // the software-pipelined instruction order costs ~40 registers (double-buffered fragments, all eight dh / hidden-tile fragments
    // live at once): taken where the instantiation stays inside its register budget without spilling (checked in the ISA)
    constexpr bool kPipe = PFA_GRAD_PIPE && (DP <= 32 || (DP == 64 && KTM <= 3));
    // FCOL: with COL the forward, too, runs the trailing observation column (k = 16 KTM, the 7x7 grid's column 48) as 32 VALU fmas on
    // the accumulators instead of a whole k-step of 8 MFMAs that multiplies three padding columns with it: KKM k-steps on MFMA
    constexpr bool FCOL = PFA_GRAD_FCOL && COL && kPipe && KKU > 4 * KTM;
    constexpr int KKM = FCOL ? 4 * KTM : KKU;
    static_assert(!PERM || (kPipe && !MH), "the permuted head layout is wired into the pipelined single-head form");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = lane_id(), wv = wave_id(), c = lane & 15, g = lane >> 4;   // (wv is a scalar: common.hpp)
    const int pair = wv & (kGradPairs - 1);
    const bool producer = wv < kGradPairs;
    float *pl = lds + pair * L::kPairFloats;
    float *w1t = lds + kGradPairs * L::kPairFloats;
    float *b1t = w1t + L::kW1Floats;
    float *w2t = b1t + L::kTabFloats;
    float *w2bt = w2t + L::kTabFloats;
    const MlpOffsets off = mlp_offsets(DP, a);

    // Build the fragment tables.  All global loads of a thread are issued before the first LDS store so they pipeline.
#ifndef PFA_DBG_SKIP_TABLES   // (timing experiments only, tools/variant_bench.py: what the table build costs per launch)
    {
        constexpr int N1 = kMT * 64 * KS / kGradThreads, N2 = kMT * 64 * 4 / kGradThreads;
        static_assert(kMT * 64 * KS % kGradThreads == 0 && kMT * 64 * 4 % kGradThreads == 0, "table sizes");
        constexpr int CH = N1 > 16 ? 16 : N1;  // W1 table in chunks of <= 16 loads per thread (wide rows: keeps the prologue small)
        static_assert(N1 % CH == 0, "table chunking");
        float tb[N2], tw[N2], tv[N2];
#pragma unroll
        for (int j = 0; j < N2; ++j) {
            const int i = threadIdx.x + j * kGradThreads;
            const int r = i & 3, ln = (i >> 2) & 63, m = i >> 8, cc = ln & 15, gg = ln >> 4;
            tb[j] = params[off.b1 + 16 * m + 4 * gg + r];
            tw[j] = w2v_at(params, off, a, slot_output(PERM, cc), 16 * m + 4 * gg + r);    // A[i=o=cc][k-slot gg] for u = 16m+4gg+r
            tv[j] = w2v_at(params, off, a, slot_output(PERM, 4 * gg + r), 16 * m + cc);    // B[k-slot gg][j=u=16m+cc] for o = 4gg+r
        }
#pragma unroll 1
        for (int j0 = 0; j0 < N1; j0 += CH) {
            float t1[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int i = threadIdx.x + (j0 + j) * kGradThreads;
                const int kk = i % KS, ln = (i / KS) & 63, m = i / (KS * 64), cc = ln & 15, gg = ln >> 4;
                t1[j] = params[off.w1 + (16 * m + cc) * DP + 4 * kk + gg];
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int i = threadIdx.x + (j0 + j) * kGradThreads;
                const int kk = i % KS, ln = (i / KS) & 63, m = i / (KS * 64);
                w1t[(m * 64 + ln) * W1S + kk] = t1[j];
            }
        }
#pragma unroll
        for (int j = 0; j < N2; ++j) {
            const int i = threadIdx.x + j * kGradThreads;
            b1t[i] = tb[j];
            w2t[i] = tw[j];
            w2bt[i] = tv[j];
        }
        if constexpr (FCOL) {   // W1[16m + 4g + r][16 KTM] in the C-fragment order of hidden^T, in the pad floats behind each lane's k-steps
            static_assert(W1S - KS >= 4 && KS % 4 == 0, "the trailing-column fragment sits in the lane's pad floats");
            float tc[N2];
#pragma unroll
            for (int j = 0; j < N2; ++j) {
                const int i = threadIdx.x + j * kGradThreads;
                const int r = i & 3, ln = (i >> 2) & 63, m = i >> 8, gg = ln >> 4;
                tc[j] = params[off.w1 + (16 * m + 4 * gg + r) * DP + 16 * KTM];
            }
#pragma unroll
            for (int j = 0; j < N2; ++j) {
                const int i = threadIdx.x + j * kGradThreads;
                w1t[(i >> 2) * W1S + KS + (i & 3)] = tc[j];
            }
        }
    }
#endif

    const long long tiles = mb_rows / 16;
    const long long pair_global = (long long)blockIdx.x * kGradPairs + pair;
    const long long pair_count = (long long)gridDim.x * kGradPairs;
    const int J = (int)((tiles + pair_count - 1) / pair_count);  // same for every pair: all waves run the same barriers
    const bool aligned = (map.horizon & 15) == 0;

    if (producer) {
        // ------------------------------------------------------------------------------------------ producer
        float bo[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bo[r] = b2v_at(params, off, a, slot_output(PERM, 4 * g + r));
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
        const float adv_rden = 1.0f / adv_den;

        f32x4 acc_dw2[kMT];
        float db2[4], stats[6];
#pragma unroll
        for (int m = 0; m < kMT; ++m) acc_dw2[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) db2[r] = 0.0f;
#pragma unroll
        for (int i = 0; i < 6; ++i) stats[i] = 0.0f;

        float4 xpre[kGlds ? 1 : NLD];  // register prefetch of the next tile's X rows (GL: none, the rows go straight to LDS)
        RowScalars rspre;
        auto prefetch = [&](long long tile, float *xdst) {   // xdst: the LDS slot of `tile` (GL)
            const bool ok = tile < tiles;
            const unsigned first = ok ? map.tile_first((unsigned)tile) : 0u;
            rspre = RowScalars{0, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (ok) {  // per-row scalars (lane c <-> row c, replicated over the 4 lane groups)
                const unsigned fr = map.tile_row((unsigned)tile, first, c, aligned);
                rspre = RowScalars{ex.actions[fr], ex.logprobs[fr], ex.values[fr], ex.advantages[fr], ex.returns[fr], 1.0f};
            }
            if constexpr (kGlds) {
                // lane l of block b fetches chunk (row, kk = 4b + (l >> 4)) with row = (l & 12) + ((l - (l >> 4)) & 3): position l of the block
                const int kl = lane >> 4, row = (lane & 12) + ((lane - kl) & 3);
                if (ok) {
                    const unsigned frow = map.tile_row((unsigned)tile, first, row, aligned);
                    const float *src = ex.obs + (size_t)frow * DP + 4 * kl;
#pragma unroll
                    for (int b = 0; b < DP / 16; ++b)
                        __builtin_amdgcn_global_load_lds(src + 16 * b, (__attribute__((address_space(3))) void *)(xdst + 256 * b), 16, 0, 0);
                } else {   // a pair's padding tile: zeros (its rows carry weight 0, but whatever the slot held must not reach the loss)
#pragma unroll
                    for (int b = 0; b < DP / 16; ++b) *reinterpret_cast<float4 *>(xdst + 256 * b + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            } else {
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const int idx = lane + 64 * j;
                xpre[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok && idx < 16 * V) {
                    const int r = idx / V, c4 = idx - r * V;
                    const unsigned row = map.tile_row((unsigned)tile, first, r, aligned);
                    xpre[j] = *reinterpret_cast<const float4 *>(ex.obs + (size_t)row * DP + 4 * c4);
                }
            }
            }
        };
        prefetch(pair_global, pl + L::kXT);
        __syncthreads();  // fragment tables ready (GL: and X(0) landed — a workgroup barrier drains the direct-to-LDS loads)

        float *hsP = pl + L::kHT, *dsP = pl + L::kDT;
        for (int j = 0; j < J; ++j) {
            float *xs = pl + L::kXT + (j % L::kXSlots) * L::kXSlot;
            PFA_STAMP(j, 0);
            // ---- stage X(j), forward, heads, loss (registers + this tile's X slot only) ------------------------------
            if constexpr (!kGlds) {
#pragma unroll
            for (int q = 0; q < NLD; ++q) {
                const int idx = lane + 64 * q;
                if (idx < 16 * V) {
                    const int r = idx / V, c4 = idx - r * V;
                    float2 *d = reinterpret_cast<float2 *>(xs + r * XS + 4 * c4);
                    d[0] = make_float2(xpre[q].x, xpre[q].y);
                    d[1] = make_float2(xpre[q].z, xpre[q].w);
                }
            }
            }
            const RowScalars rs = rspre;
            // lands during this tile's ~10k cycles (GL: in the slot X(j-2) used; the barrier pair of this tile drains it)
            prefetch(pair_global + (long long)(j + 1) * pair_count, pl + L::kXT + ((j + 1) % L::kXSlots) * L::kXSlot);
            wave_lds_fence();
            PFA_STAMP(j, 1);

            f32x4 h[kMT];
            if constexpr (kPipe) {
            // Software pipeline over the k-groups of 4 k-steps: the fragments of group k4 + 1 (8 ds_read_b128 of W1 + the X column
            // quad) are issued BEFORE the 32 MFMAs of group k4 and land under them.  Left to itself the compiler sinks every
            // ds_read next to its use and waits lgkmcnt(0) in front of each 4 MFMAs (round-2 ISA: the LDS latency exposed 26 times
            // per tile); the sched_barriers pin the order.  Within a group the products run k-step-major: 8 independent accumulator
            // chains instead of 4 dependent MFMAs on one chain.  KKU = ceil(obs_dim / 4) k-steps carry data; the rest is zero
            // padding in X and in W1 and is not issued.
            constexpr int NK4 = (ABL & 4) ? 0 : (KKM + 3) / 4;
            f32x4 wq[2][kMT];
            float xb[2][4];
            auto load_group = [&](int k4, int b) {
#pragma unroll
                for (int q = 0; q < 4; ++q) xb[b][q] = xs[L::xi(c, 4 * (4 * k4 + q) + g)];
#pragma unroll
                for (int m = 0; m < kMT; ++m) wq[b][m] = *reinterpret_cast<const f32x4 *>(w1t + (m * 64 + lane) * W1S + 4 * k4);
            };
            if (NK4 > 0) load_group(0, 0);
#pragma unroll
            for (int m = 0; m < kMT; ++m) h[m] = *reinterpret_cast<const f32x4 *>(b1t + (m * 64 + lane) * 4);
#pragma unroll
            for (int k4 = 0; k4 < NK4; ++k4) {
                if (k4 + 1 < NK4) load_group(k4 + 1, (k4 + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int m = 0; m < kMT; ++m)
                        if (4 * k4 + q < KKM) h[m] = mfma16(wq[k4 & 1][m][q], xb[k4 & 1][q], h[m]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (FCOL) {
                const float xcol = xs[L::xi(c, 16 * KTM)];     // X[row = c][16 KTM]
#pragma unroll
                for (int m = 0; m < kMT; ++m) {
                    const f32x4 wc = *reinterpret_cast<const f32x4 *>(w1t + (m * 64 + lane) * W1S + KS);
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[m][r] = fmaf(wc[r], xcol, h[m][r]);
                }
            }
            } else {
#pragma unroll
            for (int m = 0; m < kMT; ++m) h[m] = *reinterpret_cast<const f32x4 *>(b1t + (m * 64 + lane) * 4);
#pragma unroll
            for (int k4 = 0; k4 < ((ABL & 4) ? 0 : (KKU + 3) / 4); ++k4) {
                float xb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) xb[q] = xs[c * XS + 4 * (4 * k4 + q) + g];
                // KKU = ceil(obs_dim / 4) k-steps carry data; the rest is zero padding in X and in W1 and is not issued
#pragma unroll
                for (int m = 0; m < kMT; ++m) {
                    const f32x4 wq = *reinterpret_cast<const f32x4 *>(w1t + (m * 64 + lane) * W1S + 4 * k4);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (4 * k4 + q < KKU) h[m] = mfma16(wq[q], xb[q], h[m]);
                }
            }
            }
            f32x4 w4h[kMT];   // the heads' A fragments: issued here, they land under the ReLU's VALU stretch
            if constexpr (kPipe) {
#pragma unroll
                for (int m = 0; m < kMT; ++m) w4h[m] = *reinterpret_cast<const f32x4 *>(w2t + (m * 64 + lane) * 4);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < kMT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // ReLU.  fmaxf is two instructions (canonicalise + max); max(bits, 0) as integers is one and exact (negative floats
                    // are negative integers; -0 -> +0).  In the round-2 instruction order it shifted the block's schedule for the worse
                    // (66.8 vs 64.7 us); with the pipelined order it is 1 us per optimizer step faster (profiles/r03_grad_variants.txt)
                    if constexpr (kPipe) h[m][r] = __int_as_float(max(__float_as_int(h[m][r]), 0));
                    else h[m][r] = fmaxf(h[m][r], 0.0f);
                }
            f32x4 out;
            if constexpr (kPipe) {
                f32x4 o[4] = {f32x4{bo[0], bo[1], bo[2], bo[3]}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f},
                              f32x4{0.f, 0.f, 0.f, 0.f}};  // four independent chains
#pragma unroll
                for (int m0 = 0; m0 < kMT; m0 += 4)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int q = 0; q < 4; ++q) o[q] = mfma16(w4h[m0 + q][r], h[m0 + q][r], o[q]);
                out = (o[0] + o[1]) + (o[2] + o[3]);  // out^T[o = 4g + r][row = c]
            } else {
                f32x4 o[4] = {f32x4{bo[0], bo[1], bo[2], bo[3]}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f},
                              f32x4{0.f, 0.f, 0.f, 0.f}};  // four independent chains
#pragma unroll
                for (int m0 = 0; m0 < kMT; m0 += 4) {
                    f32x4 w4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) w4[q] = *reinterpret_cast<const f32x4 *>(w2t + ((m0 + q) * 64 + lane) * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int q = 0; q < 4; ++q) o[q] = mfma16(w4[q][r], h[m0 + q][r], o[q]);
                }
                out = (o[0] + o[1]) + (o[2] + o[3]);  // out^T[o = 4g + r][row = c]
            }
            if (out[0] == 12345.678f) PFA_STAMP(j, 7);   // (never true) pins the stamp behind the heads' result
            PFA_STAMP(j, 2);
            LossOut lo;
            if constexpr (ABL & 1) {
                lo.dout = out * rs.weight;
                lo.pg = lo.v_loss = lo.ent = lo.neg_logratio = lo.kl = lo.clipped = rs.adv;
            } else {
                lo = ppo_loss_tile<MH, !MH, PERM>(out, rs, a, heads, g, hp, adv_mean, adv_rden, inv_rows);   // one head: permlane-swap reductions
            }
            const f32x4 dout = lo.dout;
#pragma unroll
            for (int r = 0; r < 4; ++r) db2[r] += dout[r];
            if (g == 0) {  // one lane group owns the per-row scalars
                stats[0] += lo.pg;
                stats[1] += lo.v_loss;
                stats[2] += lo.ent;
                stats[3] += lo.neg_logratio;
                stats[4] += lo.kl;
                stats[5] += lo.clipped;
            }
            if (stats[0] == 12345.678f) PFA_STAMP(j, 7);
            PFA_STAMP(j, 3);
            __syncthreads();  // alpha_j: the consumer has finished with hidden/dout of tile j-1
            PFA_STAMP(j, 4);
            // ---- publish hidden(j) [row][u] and dout(j) [row][o] ------------------------------------------------------
#pragma unroll
            for (int m = 0; m < kMT; ++m) *reinterpret_cast<f32x4 *>(hsP + c * HS + 16 * m + 4 * g) = h[m];
            *reinterpret_cast<f32x4 *>(dsP + c * DS + 4 * g) = dout;  // dout[row = c][o = 4g..4g+3]
            __syncthreads();  // beta_j
            PFA_STAMP(j, 5);
            // ---- dW2v^T[u][o] += hidden^T . dout, both operands back from LDS in A/B fragment order -----------------------
            if constexpr (kPipe) {
            if (!(ABL & 2)) {
                float dfrag[4];  // B frags of dout[row][o] (k-slot g <-> row 4g+r, j = o = c)
                float hrow[kMT][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) dfrag[r] = dsP[(4 * g + r) * DS + c];
#pragma unroll
                for (int m = 0; m < kMT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) hrow[m][r] = hsP[(4 * g + r) * HS + 16 * m + c];  // hidden[row=4g+r][u=16m+c]
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int m = 0; m < kMT; ++m) acc_dw2[m] = mfma16(hrow[m][r], dfrag[r], acc_dw2[m]);   // 8 independent chains
            }
            } else {
            if (!(ABL & 2)) {
                float dfrag[4];  // B frags of dout[row][o] (k-slot g <-> row 4g+r, j = o = c)
#pragma unroll
                for (int r = 0; r < 4; ++r) dfrag[r] = dsP[(4 * g + r) * DS + c];
#pragma unroll
                for (int m0 = 0; m0 < kMT; m0 += 4) {  // four independent accumulator chains at a time
                    float hrow[4][4];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int r = 0; r < 4; ++r) hrow[q][r] = hsP[(4 * g + r) * HS + 16 * (m0 + q) + c];  // hidden[row=4g+r][u=16m+c]
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc_dw2[m0 + q] = mfma16(hrow[q][r], dfrag[r], acc_dw2[m0 + q]);
                }
            }
            }
        }
        __syncthreads();  // alpha_J: matches the consumer's trailing barrier pair
        __syncthreads();  // beta_J

        // ---- epilogue: producers own dW2v, db2v, stats --------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int s = 1; s < 16; s <<= 1) db2[r] += __shfl_xor(db2[r], s, 64);
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int s = 1; s < 16; s <<= 1) stats[i] += __shfl_xor(stats[i], s, 64);
        // one reduction buffer per pair where the tile / table area holds them all (kQuad: the 64-float-row instantiations), else
        // buffer 0 for the even pairs and buffer 1 for the odd ones, in turns; the sum is (p0 + p2) + (p1 + p3) either way
        float *red = lds + (kQuad ? pair : (pair & 1)) * NL::kCount;
        for (int turn = 0; turn < (kQuad ? 1 : kGradPairs / 2); ++turn) {
            if (kQuad || (pair >> 1) == turn) {
                const bool first = kQuad || turn == 0;
#pragma unroll
                for (int m = 0; m < kMT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = NL::kDw2 + (m * 4 + r) * 64 + lane;
                        red[idx] = (first ? 0.0f : red[idx]) + acc_dw2[m][r];
                    }
                if (c == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = NL::kDb2 + 4 * g + r;
                        red[idx] = (first ? 0.0f : red[idx]) + db2[r];
                    }
                }
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < kNumStats; ++i) {
                        const int idx = NL::kStats + i;
                        red[idx] = (first ? 0.0f : red[idx]) + (i < 6 ? stats[i] : 0.0f);
                    }
                }
            }
            __syncthreads();
        }
    } else {
        // ------------------------------------------------------------------------------------------ consumer
        f32x4 acc_dw1[KTM][kMT];
        float acc_col[kMT];   // COL: dW1[u = 16m + c][k = 16 KTM], per-lane partial over the lane's rows 4g + r
#pragma unroll
        for (int m = 0; m < kMT; ++m) {
#pragma unroll
            for (int kt = 0; kt < KTM; ++kt) acc_dw1[kt][m] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc_col[m] = 0.0f;
        }
        __syncthreads();  // fragment tables ready (same barrier as the producers')

        float db1[kMT];
#pragma unroll
        for (int m = 0; m < kMT; ++m) db1[m] = 0.0f;
        const float *hsP = pl + L::kHT, *dsP = pl + L::kDT;
        // The consumer's share of tile jj from the published hidden/dout tiles and the tile's X slot:
        //   dh[row][u] = dout . W2v   (A = dout[row = c][o = 4g + r], B = w2bt fragments), relu' from the hidden tile, db1,
        //   dW1^T[k][u] += X^T . dh   (A = X tile: i = k = 16kt + c, k-slot g <-> row 4g+r;  B = dh's own C fragment)
        auto backward_tile = [&](int jj) {
            const long long tile = pair_global + (long long)jj * pair_count;
            if ((ABL & 2) || tile >= tiles) return;
            const float *xs = pl + L::kXT + (jj % L::kXSlots) * L::kXSlot;
            const f32x4 dout = *reinterpret_cast<const f32x4 *>(dsP + c * DS + 4 * g);
            float xa[KTM][4], xc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int kt = 0; kt < KTM; ++kt) xa[kt][r] = xs[L::xi(4 * g + r, 16 * kt + c)];
                xc[r] = COL ? xs[L::xi(4 * g + r, 16 * KTM)] : 0.0f;
            }
            if constexpr (kPipe) {
            // every LDS read of the tile is issued up front (dout, the X fragments above, the W2v fragments, the hidden tile for
            // relu'), then three dense stretches: 32 MFMAs of dh (8 chains), the relu' / db1 / column VALU, 32 KTM MFMAs of dW1
            f32x4 dh[kMT], wb[kMT];
            float hrow[kMT][4];
#pragma unroll
            for (int m = 0; m < kMT; ++m) {
                dh[m] = f32x4{0.f, 0.f, 0.f, 0.f};
                wb[m] = *reinterpret_cast<const f32x4 *>(w2bt + (m * 64 + lane) * 4);
            }
#pragma unroll
            for (int m = 0; m < kMT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) hrow[m][r] = hsP[(4 * g + r) * HS + 16 * m + c];  // hidden[row=4g+r][u=16m+c]
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < (PERM ? 3 : 4); ++r)   // PERM: register 3 of every lane group is padding in dout and in W2v's fragments
#pragma unroll
                for (int m = 0; m < kMT; ++m) dh[m] = mfma16(dout[r], wb[m][r], dh[m]);
#pragma unroll
            for (int m = 0; m < kMT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dh[m][r] = hrow[m][r] > 0.0f ? dh[m][r] : 0.0f;  // relu'
                    db1[m] += dh[m][r];
                    if (COL) acc_col[m] = fmaf(xc[r], dh[m][r], acc_col[m]);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m = 0; m < kMT; ++m)
#pragma unroll
                    for (int kt = 0; kt < KTM; ++kt) acc_dw1[kt][m] = mfma16(xa[kt][r], dh[m][r], acc_dw1[kt][m]);
            } else {
#pragma unroll
            for (int m0 = 0; m0 < kMT; m0 += 4) {  // four hidden tiles = four independent accumulator chains
                f32x4 dh[4], wb[4];
                float hrow[4][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    dh[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    wb[q] = *reinterpret_cast<const f32x4 *>(w2bt + ((m0 + q) * 64 + lane) * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) hrow[q][r] = hsP[(4 * g + r) * HS + 16 * (m0 + q) + c];  // hidden[row=4g+r][u=16m+c]
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if constexpr (DP > 64) {
                            // Wide rows: the 32 KT dW1 accumulators fill the whole accumulation-register half of the file (256 of 512);
                            // left to the compiler these short-lived dh accumulators go there too and it spills dW1 tiles around them
                            // (round-2 ISA: 64 spilled registers).  Pinned to architectural VGPRs instead.
                            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(dh[q]) : "v"(dout[r]), "v"(wb[q][r]));
                        } else {
                            dh[q] = mfma16(dout[r], wb[q][r], dh[q]);
                        }
                    }
                if constexpr (DP > 64) {   // the inline form is invisible to the hazard recogniser: MFMA (8 passes) -> VALU read needs 11 wait states
                    asm volatile("s_nop 7\n\ts_nop 3" ::: "memory");
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dh[q][r] = hrow[q][r] > 0.0f ? dh[q][r] : 0.0f;  // relu'
                        db1[m0 + q] += dh[q][r];
                        if (COL) acc_col[m0 + q] = fmaf(xc[r], dh[q][r], acc_col[m0 + q]);
                    }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int kt = 0; kt < KTM; ++kt) acc_dw1[kt][m0 + q] = mfma16(xa[kt][r], dh[q][r], acc_dw1[kt][m0 + q]);
            }
            }
        };
        for (int j = 0; j < J; ++j) {
            PFA_STAMP(j, 0);
            if (j > 0) backward_tile(j - 1);
            if (acc_dw1[0][0][0] == 12345.678f) PFA_STAMP(j, 7);
            PFA_STAMP(j, 1);
            __syncthreads();  // alpha_j
            PFA_STAMP(j, 2);
            __syncthreads();  // beta_j
            PFA_STAMP(j, 3);
        }
        backward_tile(J - 1);
        __syncthreads();  // alpha_J / beta_J: same barrier count as the producers
        __syncthreads();

        // ---- epilogue: consumers own dW1 and db1 --------------------------------------------------------------------
#pragma unroll
        for (int m = 0; m < kMT; ++m) {
            db1[m] += __shfl_xor(db1[m], 16, 64);
            db1[m] += __shfl_xor(db1[m], 32, 64);
            if (COL) acc_col[m] = gsum<true>(acc_col[m]);   // over the lane groups: all 16 rows of the tile
        }
        float *red = lds + (kQuad ? pair : (pair & 1)) * NL::kCount;
        for (int turn = 0; turn < (kQuad ? 1 : kGradPairs / 2); ++turn) {
            if (kQuad || (pair >> 1) == turn) {
                const bool first = kQuad || turn == 0;
#pragma unroll
                for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
                    for (int m = 0; m < kMT; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int idx = NL::kDw1 + ((kt * kMT + m) * 4 + r) * 64 + lane;
                            red[idx] = (first ? 0.0f : red[idx]) + acc_dw1[kt][m][r];
                        }
                if (g == 0) {
#pragma unroll
                    for (int m = 0; m < kMT; ++m) {
                        const int idx = NL::kDb1 + 16 * m + c;
                        red[idx] = (first ? 0.0f : red[idx]) + db1[m];
                        if (COL) {   // the trailing column W1[16m + c][16 KTM]
                            const int ic = NL::kCol + 16 * m + c;
                            red[ic] = (first ? 0.0f : red[ic]) + acc_col[m];
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
#ifdef PFA_DBG_SKIP_STORE
    if (blockIdx.x != 0xFFFF) return;
#endif
    float *dst = partials + (size_t)blockIdx.x * NL::kCount;
    if constexpr (kQuad) {
        for (int i = threadIdx.x; i < NL::kCount; i += kGradThreads)
            dst[i] = (lds[i] + lds[2 * NL::kCount + i]) + (lds[NL::kCount + i] + lds[3 * NL::kCount + i]);
    } else {
        for (int i = threadIdx.x; i < NL::kCount; i += kGradThreads) dst[i] = lds[i] + lds[NL::kCount + i];
    }
}

// Fixed-order sum of the workgroup partials (native layout) + scatter into the flat gradient layout, plus one
// f64 partial of sum(g^2) per block for the gradient-norm clip.  Block = 64 native slots x 4 slices of the
// partial index; 16 independent loads in flight per thread.
#ifndef PFA_REDUCE_SLICES
#define PFA_REDUCE_SLICES 16   // slices of the partial index per slot (4 = the round-2 shape, for A/B timing)
#endif
constexpr int kRedSl = PFA_REDUCE_SLICES;
// (16 slices: 163 workgroups x 16 waves, every thread's 16 loads in flight at once — the reduction is a latency chain over the
// L2-resident partials, and 4 waves per workgroup left most SIMDs without a wave to hide it)
template <int DP, int KTM = DP / 16, bool COL = false, bool PERM = false>
__global__ void __launch_bounds__(64 * kRedSl) ppo_reduce_kernel(const float *partials, int nparts, int a, int obs_dim, float *grads,
                                                                double *norm_partials) {
    using NL = NativeLayout<DP, KTM, COL>;
    __shared__ float sh[kRedSl][64];
    __shared__ double shd[kRedSl][64];   // the loss-sum slots are carried in f64
    const int ql = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + ql;
    float acc = 0.0f;
    double dacc = 0.0;
    if (q < NL::kCount) {
        for (int i0 = sl; i0 < nparts; i0 += 16 * kRedSl) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = i0 + kRedSl * u;
                v[u] = i < nparts ? partials[(size_t)i * NL::kCount + q] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc += v[u];
                if (q >= NL::kStats) dacc += (double)v[u];
            }
        }
    }
    sh[sl][ql] = acc;
    shd[sl][ql] = dacc;
    __syncthreads();
    const MlpOffsets off = mlp_offsets(DP, a);
    if (sl == 1) {
        // W1 columns no partial slot covers (k-tiles >= KTM past the COL column: observation padding): their gradient is zero by
        // definition; written here so that the flat gradient is complete whatever the caller's buffer held
        constexpr int kFirst = 16 * KTM + (COL ? 1 : 0), kPad = DP - kFirst, kDiv = kPad > 0 ? kPad : 1;
        const int e = blockIdx.x * 64 + ql;
        if (kPad > 0 && e < kHidden * kPad) grads[off.w1 + (e / kDiv) * DP + kFirst + e % kDiv] = 0.0f;
    }
    if (sl != 0) return;
    float s = 0.0f;
    int p = -1;
    if (q < NL::kCount) {
        float t[kRedSl];
#pragma unroll
        for (int w = 0; w < kRedSl; ++w) t[w] = sh[w][ql];
#pragma unroll
        for (int w = kRedSl / 2; w > 0; w >>= 1)
#pragma unroll
            for (int x = 0; x < w; ++x) t[x] += t[x + w];   // fixed tree: deterministic, whatever the slice count
        s = t[0];
        if (COL && q >= NL::kCol && q < NL::kDw2) {
            p = off.w1 + (q - NL::kCol) * DP + 16 * KTM;
            if (16 * KTM >= obs_dim) s = 0.0f;
        } else if (q < NL::kCol) {
            const int ln = q & 63, r = (q >> 6) & 3, m = (q >> 8) & (kMT - 1), kt = q >> 11;
            p = off.w1 + (16 * m + (ln & 15)) * DP + 16 * kt + 4 * (ln >> 4) + r;
            // columns >= obs_dim are observation padding: their gradient is zero by definition (kernel A2 keeps the bias's 1.0 in the
            // first of them, which would otherwise put db1 there)
            if (16 * kt + 4 * (ln >> 4) + r >= obs_dim) s = 0.0f;
        } else if (q < NL::kDb1) {
            const int t = q - NL::kDw2, ln = t & 63, r = (t >> 6) & 3, m = t >> 8;
            const int o = slot_output(PERM, ln & 15), u = 16 * m + 4 * (ln >> 4) + r;
            if (o < a) p = off.w2 + o * kHidden + u;
            else if (o == a) p = off.wv + u;
        } else if (q < NL::kDb2) {
            p = off.b1 + (q - NL::kDb1);
        } else if (q < NL::kStats) {
            const int o = slot_output(PERM, q - NL::kDb2);
            if (o < a) p = off.b2 + o;
            else if (o == a) p = off.bv;
        } else {
            // loss sums: summed over the partials in f64 and left as (hi, lo) float pairs behind the gradient
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < kRedSl; ++w) t += shd[w][ql];
            const float hi = (float)t;
            grads[off.count + 2 * (q - NL::kStats)] = hi;
            grads[off.count + 2 * (q - NL::kStats) + 1] = (float)(t - (double)hi);
        }
        if (p >= 0) grads[p] = s;
    }
    double sq = (p >= 0 && p < off.count) ? (double)s * (double)s : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    if (ql == 0) norm_partials[blockIdx.x] = sq;
}

// torch.optim.Adam's single-tensor update of one element (clean_pufferl.py:240-244), shared by adam_clip_kernel and the one-launch
// form.  Every operation is an explicitly rounded one: left to the compiler, `a * b + c` contracts into an fma in one kernel and not
// in the other, and the two forms of the optimizer step are specified to give identical bits.  (hipcc's __fmul_rn / __fadd_rn
// wrappers contract all the same — checked in the ISA — so: plain operators with the contraction switched off for the body.)
__device__ __forceinline__ void adam_element(float &p, float &m, float &v, float g, float clip, float neg_step_size, float bc2_sqrt,
                                             float beta1, float beta2, float eps) {
#pragma clang fp contract(off)
    const float w1 = 1.0f - beta1, w2 = 1.0f - beta2;
    const float gi = g * clip;
    m = m + w1 * (gi - m);                                   // exp_avg.lerp_(grad, 1 - beta1)
    v = (v * beta2) + ((w2 * gi) * gi);                      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = (sqrtf(v) / bc2_sqrt) + eps;
    p = p + ((neg_step_size * m) / denom);                   // param.addcdiv_(exp_avg, denom, value=-step_size)
}
__device__ __forceinline__ float clip_factor(double sum_sq, float max_grad_norm) {   // clip_grad_norm_: min(1, max_norm / (norm + 1e-6))
#pragma clang fp contract(off)
    const float total_norm = (float)sqrt(sum_sq);
    if (!(max_grad_norm > 0.0f)) return 1.0f;
    const float clip = max_grad_norm / (total_norm + 1e-6f);
    return clip > 1.0f ? 1.0f : clip;
}

constexpr int kAdamThreads = 256;   // adam_clip_kernel's workgroup: its norm summation order is the contract of both forms

// ---- Kernels B + C as ONE launch: partial sums -> (data parallel: exchange with the peers) -> clip norm -> Adam ----------------
// Each workgroup owns 64 entries of the fragment-order ("native") gradient layout from the sum of the partials to the parameter
// update; the global clip norm needs every workgroup's sum of squares, so the launch carries one grid-wide barrier (all of its
// <= 300 workgroups are co-resident on 256 CUs; waves that only helped summing the partials have exited by then, a waiting
// workgroup is one wavefront).  What this saves over ppo_reduce_kernel + adam_clip_kernel: a launch (its ramp and the gap in
// front of it, ~16 times per update), the round trip of the flat gradient through L2, and — data parallel — the all-reduce as a
// launch of its own: the exchange is 8-byte {value, sequence} stores into the peers' memory and a spin on the local copy
// (p2p_ll.hpp), placed between the partial sum and the norm.  Arithmetic and summation order are those of the two-kernel form
// (single rank: bit-identical results; R ranks: every entry is the rank-order sum of the ranks' entries, identical on all of them).
// The grid-wide hand-off of the sum-of-squares pieces carries its own arrival signal: every workgroup publishes its f64 piece as
// two 8-byte words {32 bits of the value, generation of this launch} and reads everybody's words until they show this launch's
// generation — no counter, no fence (a release fence here writes back the whole L2, megabytes of the gradient kernel's partials
// still dirty in it: measured, it made the one-launch form slower than the two kernels it replaces).
struct GridWords {
    unsigned long long *words;   // [workgroups][2], library-owned, zeroed once when allocated (grid_words_of)
    unsigned gen;                // never 0, different for every launch of the process
    int *status;                 // host-pinned word: raised when the hand-off's bounded wait ran out (pfa_ppo_grid_status)
    long long timeout_ticks;     // of the 100 MHz wall clock (PFA_WAIT_TIMEOUT_MS, default 10 s)
};
struct AdamArgs {
    float *params, *exp_avg, *exp_avg_sq;
    float neg_step_size, bc2_sqrt, beta1, beta2, eps, max_grad_norm;
    double *losses;
    double loss_scale;
    // the update's LAST launch only (pfa_ppo_mlp_train_logged): the lanes that own the loss sums also leave train()'s report,
    // out10 = { losses[0..5] after this step, ev4[0..3] } — what log_pack_kernel would write in a launch of its own behind this one
    const double *log_ev4;
    double *log_out10;
};
template <int DP, int KTM, bool COL, bool PERM, bool DIST>
__global__ void __launch_bounds__(64 * kRedSl) ppo_reduce_adam_kernel(const float *partials, int nparts, int a, int obs_dim, float *grads,
                                                                     GridWords gw, AdamArgs ad, LlArgs ll) {
    using NL = NativeLayout<DP, KTM, COL>;
    __shared__ float sh[kRedSl][64];
    __shared__ double shd[kRedSl][64];   // the loss-sum slots are carried in f64
    const int ql = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + ql;
    // data parallel: one lane of the launch reads this rank's status word (host-pinned: a PCIe round trip) and publishes it to the peers
    // NOW, under the partial sums; the peers' words are collected behind the norm hand-off's publish further down (p2p_ll.hpp)
    const bool status_lane = DIST && blockIdx.x == 0 && threadIdx.x == 0;
    int status_mine = 0;
    if constexpr (DIST) {
        if (status_lane) status_mine = ll_status_push(ll);
    }
    float acc = 0.0f;
    double dacc = 0.0;
    if (q < NL::kCount) {
        for (int i0 = sl; i0 < nparts; i0 += 16 * kRedSl) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = i0 + kRedSl * u;
                v[u] = i < nparts ? partials[(size_t)i * NL::kCount + q] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc += v[u];
                if (q >= NL::kStats) dacc += (double)v[u];
            }
        }
    }
    sh[sl][ql] = acc;
    shd[sl][ql] = dacc;
    __syncthreads();
    const MlpOffsets off = mlp_offsets(DP, a);
    if (sl == 1) {   // observation-padding columns of W1 no partial slot covers: gradient zero by definition (see ppo_reduce_kernel)
        constexpr int kFirst = 16 * KTM + (COL ? 1 : 0), kPad = DP - kFirst, kDiv = kPad > 0 ? kPad : 1;
        const int e = blockIdx.x * 64 + ql;
        if (kPad > 0 && e < kHidden * kPad) grads[off.w1 + (e / kDiv) * DP + kFirst + e % kDiv] = 0.0f;
    }
    if (sl != 0) return;
    float s = 0.0f, hi = 0.0f, lo = 0.0f;
    int p = -1;
    const bool stat = q >= NL::kStats && q < NL::kCount;
    if (q < NL::kCount) {
        float t[kRedSl];
#pragma unroll
        for (int w = 0; w < kRedSl; ++w) t[w] = sh[w][ql];
#pragma unroll
        for (int w = kRedSl / 2; w > 0; w >>= 1)
#pragma unroll
            for (int x = 0; x < w; ++x) t[x] += t[x + w];
        s = t[0];
        if (COL && q >= NL::kCol && q < NL::kDw2) {
            p = off.w1 + (q - NL::kCol) * DP + 16 * KTM;
            if (16 * KTM >= obs_dim) s = 0.0f;
        } else if (q < NL::kCol) {
            const int ln = q & 63, r = (q >> 6) & 3, m = (q >> 8) & (kMT - 1), kt = q >> 11;
            p = off.w1 + (16 * m + (ln & 15)) * DP + 16 * kt + 4 * (ln >> 4) + r;
            if (16 * kt + 4 * (ln >> 4) + r >= obs_dim) s = 0.0f;
        } else if (q < NL::kDb1) {
            const int t2 = q - NL::kDw2, ln = t2 & 63, r = (t2 >> 6) & 3, m = t2 >> 8;
            const int o = slot_output(PERM, ln & 15), u = 16 * m + 4 * (ln >> 4) + r;
            if (o < a) p = off.w2 + o * kHidden + u;
            else if (o == a) p = off.wv + u;
        } else if (q < NL::kDb2) {
            p = off.b1 + (q - NL::kDb1);
        } else if (q < NL::kStats) {
            const int o = slot_output(PERM, q - NL::kDb2);
            if (o < a) p = off.b2 + o;
            else if (o == a) p = off.bv;
        } else {
            double t2 = 0.0;
#pragma unroll
            for (int w = 0; w < kRedSl; ++w) t2 += shd[w][ql];
            hi = (float)t2;                       // the (hi, lo) float pair of the f64 sum, as the two-kernel form leaves it
            lo = (float)(t2 - (double)hi);
        }
    }
    // this entry's parameter and moments: in flight across the exchange and the barrier
    float p_i = 0.0f, m_i = 0.0f, v_i = 0.0f;
    if (p >= 0) {
        p_i = ad.params[p];
        m_i = ad.exp_avg[p];
        v_i = ad.exp_avg_sq[p];
    }
    if constexpr (DIST) {   // one hop: my entry into every peer's memory, theirs out of mine, summed in rank order
        if (p >= 0) ll_push(ll, (unsigned)q, s);
        if (stat) {
            ll_push(ll, (unsigned)(NL::kStats + 2 * (q - NL::kStats)), hi);
            ll_push(ll, (unsigned)(NL::kStats + 2 * (q - NL::kStats) + 1), lo);
        }
        long long waited = 0;
        if (p >= 0) s = ll_wait_sum(ll, (unsigned)q, s, &waited);
        if (stat) {
            hi = ll_wait_sum(ll, (unsigned)(NL::kStats + 2 * (q - NL::kStats)), hi, &waited);
            lo = ll_wait_sum(ll, (unsigned)(NL::kStats + 2 * (q - NL::kStats) + 1), lo, &waited);
        }
        ll_wait_report(ll, waited);
    }
    if (p >= 0) grads[p] = s;
    if (stat) {
        grads[off.count + 2 * (q - NL::kStats)] = hi;
        grads[off.count + 2 * (q - NL::kStats) + 1] = lo;
    }
    double sq = (p >= 0 && p < off.count) ? (double)s * (double)s : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    // ---- grid-wide hand-off: publish this workgroup's piece of sum(g^2), collect everyone's ----
    if (ql < 2) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(sq);
        const unsigned half = ql == 0 ? (unsigned)bits : (unsigned)(bits >> 32);
        __hip_atomic_store(gw.words + 2 * blockIdx.x + ql, ((unsigned long long)gw.gen << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if constexpr (DIST) {   // the ranks' status words ride the exchange: this workgroup's norm piece is out, nobody waits for this lane
        if (status_lane) ll_status_wait(ll, status_mine);
    }
    // the norm in adam_clip_kernel's order: 256 strided partial sums, a butterfly per 64, then the four in order
    const int npart = (int)gridDim.x;
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < kAdamThreads / 64; ++w) {
        double ss = 0.0;
        for (int i = 64 * w + ql; i < npart; i += kAdamThreads) {
            // The wait is bounded (advisor, round 4): the hand-off relies on every workgroup of the launch being resident at once.
            // The host checks that once per shape (reduce_adam_coresident: occupancy x CUs >= workgroups, else the two-kernel
            // form runs), but a device shared with other processes or masked down to fewer CUs can still starve a workgroup; then
            // the wait runs out, the status word is raised (clean_pufferl.train raises) and the norm becomes NaN, which poisons the
            // parameters instead of hanging the GPU.
            unsigned long long w0, w1;
            long long t0 = 0;
            int spin = 0;
            bool lost = false;
            while (true) {
                w0 = __hip_atomic_load(gw.words + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                w1 = __hip_atomic_load(gw.words + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(w0 >> 32) == gw.gen && (unsigned)(w1 >> 32) == gw.gen) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spin & 255) == 0) {
                    const long long now = (long long)wall_clock64();
                    if (t0 == 0) t0 = now;
                    else if (now - t0 > gw.timeout_ticks) {
                        lost = true;
                        break;
                    }
                }
            }
            if (lost) {
                __hip_atomic_store(gw.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                ss = __builtin_nan("");
                break;
            }
            ss += __longlong_as_double((long long)(((w1 & 0xffffffffull) << 32) | (w0 & 0xffffffffull)));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        tot += ss;
    }
    const float clip = clip_factor(tot, ad.max_grad_norm);
    if (p >= 0) {
        adam_element(p_i, m_i, v_i, s, clip, ad.neg_step_size, ad.bc2_sqrt, ad.beta1, ad.beta2, ad.eps);
        ad.params[p] = p_i;
        ad.exp_avg[p] = m_i;
        ad.exp_avg_sq[p] = v_i;
    }
    if (stat) {
        const int i = q - NL::kStats;
        double run = 0.0;
        if (ad.losses && i < 6) {
            run = ad.losses[i] + ((double)hi + (double)lo) * ad.loss_scale;
            ad.losses[i] = run;
        }
        if (ad.log_out10) {
            static_assert(kNumStats == 8, "slots 6 and 7 carry the four explained-variance sums of the report");
            if (i < 6) {
                ad.log_out10[i] = run;
            } else {
                ad.log_out10[6 + 2 * (i - 6)] = ad.log_ev4[2 * (i - 6)];
                ad.log_out10[7 + 2 * (i - 6)] = ad.log_ev4[2 * (i - 6) + 1];
            }
        }
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

// Sums behind the reference's explained-variance log line (clean_pufferl.py:266-270):
//   y_pred = values in STORAGE (step-major) order, y_true = advantages (env-major order) + y_pred  (mis-aligned on
//   purpose, it only feeds logging).  out[0..3] = sum y_true, sum y_true^2, sum adv, sum adv^2 in f64.
constexpr int kEvBlocks = 128;
__global__ void __launch_bounds__(256) ev_partial_kernel(const float *adv, const float *values, long long n, int num_envs,
                                                        int horizon, double *partial /* [kEvBlocks][4] */) {
    __shared__ double sh[4][256];
    double a[4] = {0, 0, 0, 0};
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < n; j += (long long)kEvBlocks * 256) {
        const long long e = j % num_envs, t = j / num_envs;  // storage order index j = t*N + e
        const double yp = (double)values[e * horizon + t];
        const double ad = (double)adv[j];
        const double yt = ad + yp;
        a[0] += yt;
        a[1] += yt * yt;
        a[2] += ad;
        a[3] += ad * ad;
    }
    for (int q = 0; q < 4; ++q) sh[q][threadIdx.x] = a[q];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int q = 0; q < 4; ++q) sh[q][threadIdx.x] += sh[q][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 4) partial[blockIdx.x * 4 + threadIdx.x] = sh[threadIdx.x][0];
}
// out[0..5] = the six running loss means (f32 -> f64), out[6..9] = the four sums: one D2H copy serves train()'s log line.
__global__ void ev_final_kernel(const double *partial, const double *losses, double *out10) {
    const int q = threadIdx.x;
    if (q < 6) out10[q] = losses ? losses[q] : 0.0;
    if (q >= 6 && q < 10) {
        double s = 0.0;
        for (int b = 0; b < kEvBlocks; ++b) s += partial[b * 4 + (q - 6)];
        out10[q] = s;
    }
}

// clip_grad_norm_ + torch.optim.Adam (single-tensor path) on the flat parameter vector.
// Every workgroup recomputes the global gradient norm from the (L2-resident, ~38 KB) gradient in the same fixed
// order — deterministic and no grid-wide hand-off — then updates its own 256-element slice.
__global__ void __launch_bounds__(kAdamThreads) adam_clip_kernel(float *params, const float *grads, float *exp_avg,
                                                                float *exp_avg_sq, long long count, float neg_step_size,
                                                                float bc2_sqrt, float beta1, float beta2, float eps,
                                                                float max_grad_norm, float grad_scale, const float *loss_sums,
                                                                double *losses, double loss_scale, const double *norm_partials,
                                                                int n_norm_partials) {
    __shared__ double sh[kAdamThreads / 64];
    // this thread's element: loaded BEFORE the norm reduction, so the two memory latencies overlap instead of adding up
    const long long i = (long long)blockIdx.x * kAdamThreads + threadIdx.x;
    float g_i = 0.0f, m = 0.0f, v = 0.0f, p_i = 0.0f;
    if (i < count) {
        g_i = grads[i];
        m = exp_avg[i];
        v = exp_avg_sq[i];
        p_i = params[i];
    }
    double ss = 0.0;
    if (norm_partials) {  // sum(g^2) pieces left by ppo_reduce_kernel (single rank: no all-reduce in between)
        for (int i = threadIdx.x; i < n_norm_partials; i += kAdamThreads) ss += norm_partials[i];
        ss *= (double)grad_scale * (double)grad_scale;
    } else {
        for (long long base = threadIdx.x; base < count; base += kAdamThreads * 8) {
            float gv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long long i = base + (long long)u * kAdamThreads;
                gv[u] = i < count ? grads[i] * grad_scale : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) ss += (double)gv[u] * (double)gv[u];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if (lane_id() == 0) sh[wave_id()] = ss;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < kAdamThreads / 64; ++w) tot += sh[w];
    const float clip = clip_factor(tot, max_grad_norm);
    if (i < count) {
        adam_element(p_i, m, v, g_i * grad_scale, clip, neg_step_size, bc2_sqrt, beta1, beta2, eps);
        params[i] = p_i;
        exp_avg[i] = m;
        exp_avg_sq[i] = v;
    }
    if (blockIdx.x == 0 && loss_sums && losses && threadIdx.x < 6)   // (hi, lo) pairs of the f64 sums
        losses[threadIdx.x] += ((double)loss_sums[2 * threadIdx.x] + (double)loss_sums[2 * threadIdx.x + 1]) * loss_scale;
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

// the 7x7 grid behind one Discrete head on 64-float rows runs the instantiation with 3 dW1 k-tiles + the column-48 accumulator
static bool grad_trimmed(const pfa_mlp_dims *dims) { return dims->obs_stride == 64 && !dims->heads && dims->obs_dim == 49; }
#ifndef PFA_GRAD_PERM
#define PFA_GRAD_PERM 1   // 0 = head outputs in natural fragment rows everywhere (A/B timing)
#endif
// ... and, with at most 11 actions (12 outputs with the value), the permuted head rows: three k-steps of dh instead of four
static bool grad_perm(const pfa_mlp_dims *dims) { return PFA_GRAD_PERM && dims->num_actions <= 11; }
// the opt-in product form (pfa_igemm_set_products(1): every fp32 product as six bf16 partial products, fp32 accumulate) covers the
// headline shape: the 7x7 grid on 64-float rows behind one Discrete head, minibatches of whole 32-row tiles
static bool grad_bf16(const pfa_mlp_dims *dims, int64_t mbs) { return pfa_igemm_get_products() == 1 && grad_trimmed(dims) && mbs >= 32 && mbs % 32 == 0; }
static int norm_blocks(const pfa_mlp_dims *dims) {   // workgroups of the reduce launch = f64 pieces of sum(g^2) it leaves
    const int count = grad_trimmed(dims) ? NativeLayout<64, 3, true>::kCount
                                         : (dims->obs_stride / 16) * kMT * 4 * 64 + kMT * 4 * 64 + kHidden + kOut + kNumStats;
    return (count + 63) / 64;
}
static size_t native_count(int dp) { return (size_t)(dp / 16) * kMT * 4 * 64 + kMT * 4 * 64 + kHidden + kOut + kNumStats; }
static size_t partials_bytes(const pfa_mlp_dims *dims) {   // (the opt-in bf16 form runs two workgroups per CU: kBfMaxGrid partials)
    return align_up((size_t)(dims->obs_stride == 64 ? kBfMaxGrid : 256) * native_count(dims->obs_stride) * sizeof(float), 256);
}
static double *norm_partials_of(void *workspace, const pfa_mlp_dims *dims) {  // after the gradient partials
    return (double *)((char *)workspace + partials_bytes(dims));
}
// The grid hand-off words are LIBRARY-OWNED (advisor, round 5: they used to live at the end of the caller's workspace, whose contract then
// silently was "zeroed once and never written by anybody else"): one small device buffer per workspace pointer, allocated and cleared
// on first use (grid_words_of below).  A word only ever holds {value bits, generation of the launch that wrote it}, generations never
// repeat within a process and generation 0 is never used — so the buffer needs no clearing between launches, and a recycled or
// uninitialised caller workspace cannot pass for a hand-off.
constexpr size_t kGridWordsBytes = 16384;   // 2 words per workgroup of the reduce + Adam launch (<= 600 workgroups)
static size_t tail_bytes(const pfa_mlp_dims *dims, int nmb) {   // what follows the partials: adv-stat partials / norm pieces (shared)
    const size_t native = native_count(dims->obs_stride);
    const size_t advp = align_up((size_t)nmb * 64 * 2 * sizeof(double), 256);
    const size_t normp = align_up(((native + 63) / 64) * 2 * sizeof(double), 256);   // (the one-launch form: two 8-byte words per piece)
    return advp > normp ? advp : normp;
}


static int grad_grid(int64_t mb_rows, int dp) {
    const int64_t tiles = mb_rows / 16;
    const int64_t wgs = (tiles + grad_pairs(dp) - 1) / grad_pairs(dp);
    return (int)(wgs < 256 ? (wgs < 1 ? 1 : wgs) : 256);
}

}  // namespace pfa

using namespace pfa;

extern "C" size_t pfa_ppo_workspace_bytes(const pfa_mlp_dims *dims, int64_t batch_rows, const pfa_ppo_hparams *hp) {
    if (!dims || !hp || hp->num_minibatches < 1) return 0;
    (void)batch_rows;
    static_assert(kAdvChunks == 64, "tail_bytes assumes 64 chunks");
    // [gradient partials | adv-stat partials / norm pieces (shared)]
    return partials_bytes(dims) + tail_bytes(dims, hp->num_minibatches) + 256;
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

// Kernel A of one optimizer step: validates the arguments and launches the fused forward + loss + backward over minibatch `mb`;
// leaves *grid_out workgroup partials in the workspace.
static int launch_grad(const pfa_experience *exp, int64_t batch_rows, int32_t mb, const float *params,
                       const pfa_mlp_dims *dims, const pfa_ppo_hparams *hp, const double *adv_stats,
                       int64_t global_mb_rows, float *grads, void *workspace, pfa_stream_t stream, int *grid_out) {
    if (int rc = check_update_args(exp, batch_rows, dims, hp)) return rc;
    PFA_REQUIRE(dims && dims->hidden == kHidden, "ppo.grad: hidden must be %d", kHidden);
    PFA_REQUIRE(dims->obs_stride == 16 || dims->obs_stride == 32 || dims->obs_stride == 64 || dims->obs_stride == 96 ||
                    dims->obs_stride == 128,
                "ppo.grad: obs_stride must be 16/32/64/96/128");
    PFA_REQUIRE(dims->num_actions >= 1 && dims->num_actions <= 15, "ppo.grad: num_actions must be in 1..15");
    PFA_REQUIRE(dims->heads == 0 || heads_count(dims->heads, dims->num_actions) >= 1, "ppo.grad: head sizes 0x%x do not sum to num_actions %d",
                dims->heads, dims->num_actions);
    PFA_REQUIRE(mb >= 0 && mb < hp->num_minibatches, "ppo.grad: minibatch index out of range");
    PFA_REQUIRE(exp->obs && exp->actions && exp->logprobs && exp->values && exp->advantages && exp->returns && params && grads &&
                    workspace,
                "ppo.grad: null buffer");
    PFA_REQUIRE(!hp->norm_adv || adv_stats, "ppo.grad: norm_adv needs adv_stats");
    const int64_t mbs = batch_rows / hp->num_minibatches;
    PFA_REQUIRE(mbs % 16 == 0, "ppo.grad: minibatch_size must be a multiple of 16 (got %lld)", (long long)mbs);
    PFA_REQUIRE(global_mb_rows >= mbs, "ppo.grad: global_mb_rows < local minibatch rows");
    RowMap map{mb, hp->num_minibatches, hp->bptt_horizon};
    float *partials = (float *)workspace;
    if (grad_bf16(dims, mbs)) {   // opt-in product form (pfa_igemm_set_products(1)): the same step on the bf16 matrix path, csrc/ppo_bf16.hpp
        const int64_t tiles32 = mbs / 32;
        const int grid = (int)(tiles32 < kBfMaxGrid ? tiles32 : kBfMaxGrid);
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        const bool timed = timing_pair("ppo_mlp_grad", &ev0, &ev1);
        ScopedKernelTimer timer(timing_ext_mode() ? nullptr : "ppo_mlp_grad", (hipStream_t)stream);
#define PFA_LAUNCH_GRAD_BF16(PERMV)                                                                                                       \
    {                                                                                                                                    \
        static bool attr_set = false;                                                                                                    \
        if (!attr_set) {                                                                                                                 \
            PFA_CHECK_HIP(hipFuncSetAttribute((const void *)ppo_mlp_grad_bf16_kernel<PERMV>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                              (int)BfLds::kBytes));                                                                      \
            attr_set = true;                                                                                                             \
        }                                                                                                                                \
        if (timed)                                                                                                                       \
            hipExtLaunchKernelGGL((ppo_mlp_grad_bf16_kernel<PERMV>), dim3(grid), dim3(kBfThreads), BfLds::kBytes, (hipStream_t)stream,    \
                                  ev0, ev1, 0, *exp, map, (long long)mbs, params, dims->num_actions, *hp, adv_stats,                     \
                                  (double)global_mb_rows, partials);                                                                     \
        else                                                                                                                             \
            hipLaunchKernelGGL((ppo_mlp_grad_bf16_kernel<PERMV>), dim3(grid), dim3(kBfThreads), BfLds::kBytes, (hipStream_t)stream, *exp, \
                               map, (long long)mbs, params, dims->num_actions, *hp, adv_stats, (double)global_mb_rows, partials);        \
    }
        if (grad_perm(dims)) PFA_LAUNCH_GRAD_BF16(true)
        else PFA_LAUNCH_GRAD_BF16(false)
#undef PFA_LAUNCH_GRAD_BF16
        PFA_LAUNCH_CHECK();
        *grid_out = grid;
        return 0;
    }
    const int grid = grad_grid(mbs, dims->obs_stride);
#define PFA_LAUNCH_GRAD_FULL(DPV, KKUV, MHV, KTMV, COLV) PFA_LAUNCH_GRAD_PERM(DPV, KKUV, MHV, KTMV, COLV, false)
#define PFA_LAUNCH_GRAD_PERM(DPV, KKUV, MHV, KTMV, COLV, PERMV)                                                             \
    {                                                                                                                      \
        constexpr size_t lds_bytes = (size_t)GradLds<DPV, grad_glds(DPV, KTMV)>::kFloats * sizeof(float);                       \
        static_assert((size_t)2 * NativeLayout<DPV, KTMV, COLV>::kCount * sizeof(float) <= lds_bytes,                       \
                      "the two reduction buffers must fit in the tile/table area");                                       \
        static bool attr_set = false;                                                                                      \
        if (!attr_set) {                                                                                                   \
            PFA_CHECK_HIP(hipFuncSetAttribute((const void *)ppo_mlp_grad_kernel<DPV, 0, KKUV, MHV, KTMV, COLV, PERMV>,     \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));                \
            attr_set = true;                                                                                               \
        }                                                                                                                  \
        if (grad_timed)                                                                                                    \
            hipExtLaunchKernelGGL((ppo_mlp_grad_kernel<DPV, 0, KKUV, MHV, KTMV, COLV, PERMV>), dim3(grid), dim3(grad_threads(DPV)), \
                                  lds_bytes, (hipStream_t)stream, grad_ev0, grad_ev1, 0, *exp, map, (long long)mbs, params,  \
                                  dims->num_actions, dims->heads, *hp, adv_stats, (double)global_mb_rows, partials);      \
        else                                                                                                               \
            hipLaunchKernelGGL((ppo_mlp_grad_kernel<DPV, 0, KKUV, MHV, KTMV, COLV, PERMV>), dim3(grid), dim3(grad_threads(DPV)), \
                               lds_bytes, (hipStream_t)stream, *exp, map, (long long)mbs, params, dims->num_actions,       \
                               dims->heads, *hp, adv_stats, (double)global_mb_rows, partials);                             \
    }
#define PFA_LAUNCH_GRAD_KM(DPV, KKUV, MHV) PFA_LAUNCH_GRAD_FULL(DPV, KKUV, MHV, DPV / 16, false)
#define PFA_LAUNCH_GRAD_K(DPV, KKUV) PFA_LAUNCH_GRAD_KM(DPV, KKUV, false)
#define PFA_LAUNCH_GRAD(DPV)                                        \
    if (dims->heads) PFA_LAUNCH_GRAD_KM(DPV, DPV / 4, true)         \
    else PFA_LAUNCH_GRAD_KM(DPV, DPV / 4, false)
    {
        // bench.py's roofline leg: the launch carries its own events (stamped with the dispatch's begin and end); with
        hipEvent_t grad_ev0 = nullptr, grad_ev1 = nullptr;
        const bool grad_timed = timing_pair("ppo_mlp_grad", &grad_ev0, &grad_ev1);
        ScopedKernelTimer timer(timing_ext_mode() ? nullptr : "ppo_mlp_grad", (hipStream_t)stream);
        switch (dims->obs_stride) {
            case 16: PFA_LAUNCH_GRAD(16) break;
            case 32: PFA_LAUNCH_GRAD(32) break;
            case 96: PFA_LAUNCH_GRAD(96) break;
            case 128: PFA_LAUNCH_GRAD(128) break;
            default:
                if (grad_trimmed(dims) && grad_perm(dims)) PFA_LAUNCH_GRAD_PERM(64, 13, false, 3, true, true)   // + the permuted head rows (<= 11 actions)
                else if (grad_trimmed(dims)) PFA_LAUNCH_GRAD_FULL(64, 13, false, 3, true)   // 7x7 grid: 13 of 16 forward k-steps, dW1 = 3 k-tiles + column 48
                else if (!dims->heads && (dims->obs_dim + 3) / 4 == 13) PFA_LAUNCH_GRAD_K(64, 13)
                else PFA_LAUNCH_GRAD(64)
                break;
        }
    }
#undef PFA_LAUNCH_GRAD_KM
#undef PFA_LAUNCH_GRAD_FULL
#undef PFA_LAUNCH_GRAD_PERM
#undef PFA_LAUNCH_GRAD
#undef PFA_LAUNCH_GRAD_K
    PFA_LAUNCH_CHECK();
    *grid_out = grid;
    return 0;
}

extern "C" int pfa_ppo_mlp_grad(const pfa_experience *exp, int64_t batch_rows, int32_t mb, const float *params,
                                const pfa_mlp_dims *dims, const pfa_ppo_hparams *hp, const double *adv_stats,
                                int64_t global_mb_rows, float *grads, void *workspace, pfa_stream_t stream) {
    int grid = 0;
    if (int rc = launch_grad(exp, batch_rows, mb, params, dims, hp, adv_stats, global_mb_rows, grads, workspace, stream, &grid)) return rc;
    float *partials = (float *)workspace;
    double *normp = norm_partials_of(workspace, dims);
    ScopedKernelTimer timer2("ppo_reduce", (hipStream_t)stream);
#define PFA_LAUNCH_REDUCE(DPV, KTMV, COLV) PFA_LAUNCH_REDUCE_P(DPV, KTMV, COLV, false)
#define PFA_LAUNCH_REDUCE_P(DPV, KTMV, COLV, PERMV)                                                                             \
    hipLaunchKernelGGL((ppo_reduce_kernel<DPV, KTMV, COLV, PERMV>), dim3((NativeLayout<DPV, KTMV, COLV>::kCount + 63) / 64),     \
                       dim3(64 * kRedSl), 0, (hipStream_t)stream, partials, grid, dims->num_actions, dims->obs_dim, grads, normp)
    switch (dims->obs_stride) {   // the same instantiation choice as the gradient launch above: the partial layout belongs to it
        case 16: PFA_LAUNCH_REDUCE(16, 1, false); break;
        case 32: PFA_LAUNCH_REDUCE(32, 2, false); break;
        case 96: PFA_LAUNCH_REDUCE(96, 6, false); break;
        case 128: PFA_LAUNCH_REDUCE(128, 8, false); break;
        default:
            if (grad_trimmed(dims) && grad_perm(dims)) PFA_LAUNCH_REDUCE_P(64, 3, true, true);
            else if (grad_trimmed(dims)) PFA_LAUNCH_REDUCE(64, 3, true);
            else PFA_LAUNCH_REDUCE(64, 4, false);
            break;
    }
#undef PFA_LAUNCH_REDUCE
#undef PFA_LAUNCH_REDUCE_P
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_ppo_mlp_grad_path(const pfa_mlp_dims *dims, int64_t mb_rows) { return dims && grad_bf16(dims, mb_rows) ? 1 : 0; }

// MFMA instructions per 16-row tile of the instantiation pfa_ppo_mlp_grad dispatches for these dimensions (see the switch there):
// forward KKU x kMT, heads 4 x kMT, dW2v 4 x kMT, dh 4 x kMT, dW1 KTM x 4 x kMT.
extern "C" int pfa_ppo_mlp_grad_mfma_per_tile(int32_t obs_dim, int32_t obs_stride, int32_t num_actions) {
    if (num_actions < 1 || num_actions > 15) return 0;
    if (obs_stride != 16 && obs_stride != 32 && obs_stride != 64 && obs_stride != 96 && obs_stride != 128) return 0;
    int kku = obs_stride / 4, ktm = obs_stride / 16;
    if (obs_stride == 64 && (obs_dim + 3) / 4 == 13) kku = 13;
    if (obs_stride == 64 && obs_dim == 49) ktm = 3;
    if (PFA_GRAD_FCOL && obs_stride == 64 && obs_dim == 49) kku = 12;   // the forward's column 48 runs on the VALU (FCOL)
    const bool perm = PFA_GRAD_PERM && obs_stride == 64 && obs_dim == 49 && num_actions <= 11;   // (one Discrete head assumed, as everywhere in this helper)
    return kku * kMT + (perm ? 11 : 12) * kMT + ktm * 4 * kMT;
}

extern "C" int pfa_adam_clip_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t count, float lr,
                                  float beta1, float beta2, float eps, int64_t step, float max_grad_norm, float grad_scale,
                                  const float *loss_sums, double *losses, double loss_scale, const double *norm_partials,
                                  int32_t n_norm_partials, pfa_stream_t stream) {
    PFA_REQUIRE(params && grads && exp_avg && exp_avg_sq, "adam: null buffer");
    PFA_REQUIRE(count >= 1 && step >= 1, "adam: count and step must be >= 1");
    // torch.optim.Adam (single tensor): bias corrections and step size are python floats (f64)
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
    const float neg_step_size = (float)(-(double)lr / bc1);
    const float bc2_sqrt = (float)std::sqrt(bc2);
    ScopedKernelTimer timer("adam_clip", (hipStream_t)stream);
    hipLaunchKernelGGL(adam_clip_kernel, dim3((unsigned)((count + kAdamThreads - 1) / kAdamThreads)), dim3(kAdamThreads), 0,
                       (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, (long long)count, neg_step_size, bc2_sqrt, beta1,
                       beta2, eps, max_grad_norm, grad_scale, loss_sums, losses, loss_scale, norm_partials, (int)n_norm_partials);
    PFA_LAUNCH_CHECK();
    return 0;
}

// Kernels B + C in one launch (ppo_reduce_adam_kernel); `ll` non-null: with the data-parallel exchange inside.
static unsigned long long *grid_words_of(void *workspace) {   // one library-owned, zero-initialised buffer per trainer (keyed on its workspace)
    static std::unordered_map<void *, unsigned long long *> words;
    auto it = words.find(workspace);
    if (it != words.end()) return it->second;
    if (words.size() >= 64) {   // trainers come and go: start over (nothing of a finished launch lives in the words)
        (void)hipDeviceSynchronize();
        for (auto &kv : words) (void)hipFree(kv.second);
        words.clear();
    }
    void *p = nullptr;
    if (hipMalloc(&p, kGridWordsBytes) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, kGridWordsBytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(p);
        return nullptr;
    }
    words[workspace] = (unsigned long long *)p;
    return (unsigned long long *)p;
}
static int *grid_status_word() {   // host-pinned, device-visible, one per process
    static int *word = nullptr;
    if (!word) {
        void *p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocMapped) != hipSuccess) return nullptr;
        *(volatile int *)p = 0;
        word = (int *)p;
    }
    return word;
}
static long long grid_timeout_ticks() {
    static const long long ticks = [] {
        long long ms = 10000;
        if (const char *e = std::getenv("PFA_WAIT_TIMEOUT_MS")) {
            const long long v = std::atoll(e);
            if (v > 0) ms = v;
        }
        return ms * 100000;   // wall_clock64: 100 MHz
    }();
    return ticks;
}
// The one-launch form needs all of its workgroups resident at once (its grid-wide hand-off spins on the others' words): checked
// against the occupancy the runtime reports for the instantiation this shape launches, once per shape; otherwise (a smaller part, a
// CU mask) pfa_ppo_mlp_train runs the two-kernel form, which has no such requirement.
template <typename K>
static bool coresident(K kernel, int blocks) {
    int per_cu = 0, cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)kernel, 64 * kRedSl, 0) != hipSuccess) return false;
    return (long long)per_cu * cus >= blocks;
}
static bool reduce_adam_coresident(const pfa_mlp_dims *dims, bool dist) {
    static int cache[2][8] = {};   // [dist][shape slot]: 0 unknown, 1 yes, 2 no
    const int slot = dims->obs_stride == 16 ? 0 : dims->obs_stride == 32 ? 1 : dims->obs_stride == 96 ? 2 : dims->obs_stride == 128 ? 3
                   : (grad_trimmed(dims) && grad_perm(dims)) ? 4 : grad_trimmed(dims) ? 5 : 6;
    int &c = cache[dist ? 1 : 0][slot];
    if (c) return c == 1;
    const int blocks = norm_blocks(dims);
    bool ok;
#define PFA_CO(DPV, KTMV, COLV, PERMV) \
    ok = dist ? coresident(ppo_reduce_adam_kernel<DPV, KTMV, COLV, PERMV, true>, blocks) : coresident(ppo_reduce_adam_kernel<DPV, KTMV, COLV, PERMV, false>, blocks)
    switch (slot) {
        case 0: PFA_CO(16, 1, false, false); break;
        case 1: PFA_CO(32, 2, false, false); break;
        case 2: PFA_CO(96, 6, false, false); break;
        case 3: PFA_CO(128, 8, false, false); break;
        case 4: PFA_CO(64, 3, true, true); break;
        case 5: PFA_CO(64, 3, true, false); break;
        default: PFA_CO(64, 4, false, false); break;
    }
#undef PFA_CO
    c = ok ? 1 : 2;
    return ok;
}
static int launch_reduce_adam(const pfa_mlp_dims *dims, int nmb, int grid, float *params, float *grads, float *exp_avg, float *exp_avg_sq,
                              float lr, float beta1, float beta2, float eps, int64_t step, float max_grad_norm, double *losses,
                              double loss_scale, void *workspace, const LlArgs *ll, hipStream_t stream, const double *log_ev4 = nullptr,
                              double *log_out10 = nullptr) {
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
    AdamArgs ad{params, exp_avg, exp_avg_sq, (float)(-(double)lr / bc1), (float)std::sqrt(bc2), beta1, beta2, eps, max_grad_norm, losses, loss_scale,
                log_ev4, log_out10};
    const float *partials = (const float *)workspace;
    static unsigned long long launches = 0;
    unsigned gen = (unsigned)(++launches);
    if (gen == 0) gen = (unsigned)(++launches);
    int *gstatus = grid_status_word();
    PFA_REQUIRE(gstatus, "ppo.train: cannot allocate the grid hand-off's status word");
    unsigned long long *gwords = grid_words_of(workspace);
    PFA_REQUIRE(gwords, "ppo.train: cannot allocate the grid hand-off words");
    PFA_REQUIRE((size_t)((native_count(dims->obs_stride) + 63) / 64) * 2 * sizeof(unsigned long long) <= kGridWordsBytes, "ppo.train: hand-off words too small");
    (void)nmb;
    const GridWords gw{gwords, gen, gstatus, grid_timeout_ticks()};
    const LlArgs none{};
    ScopedKernelTimer timer("ppo_reduce_adam", stream);
#define PFA_LAUNCH_RA(DPV, KTMV, COLV, PERMV)                                                                                        \
    {                                                                                                                               \
        const dim3 g((NativeLayout<DPV, KTMV, COLV>::kCount + 63) / 64), b(64 * kRedSl);                                            \
        if (ll) hipLaunchKernelGGL((ppo_reduce_adam_kernel<DPV, KTMV, COLV, PERMV, true>), g, b, 0, stream, partials, grid,          \
                                   dims->num_actions, dims->obs_dim, grads, gw, ad, *ll);                                   \
        else hipLaunchKernelGGL((ppo_reduce_adam_kernel<DPV, KTMV, COLV, PERMV, false>), g, b, 0, stream, partials, grid,            \
                                dims->num_actions, dims->obs_dim, grads, gw, ad, none);                                     \
    }
    switch (dims->obs_stride) {   // the instantiation choice of the gradient launch: the partial layout belongs to it
        case 16: PFA_LAUNCH_RA(16, 1, false, false) break;
        case 32: PFA_LAUNCH_RA(32, 2, false, false) break;
        case 96: PFA_LAUNCH_RA(96, 6, false, false) break;
        case 128: PFA_LAUNCH_RA(128, 8, false, false) break;
        default:
            if (grad_trimmed(dims) && grad_perm(dims)) PFA_LAUNCH_RA(64, 3, true, true)
            else if (grad_trimmed(dims)) PFA_LAUNCH_RA(64, 3, true, false)
            else PFA_LAUNCH_RA(64, 4, false, false)
            break;
    }
#undef PFA_LAUNCH_RA
    PFA_LAUNCH_CHECK();
    return 0;
}
static size_t ll_entries_needed(const pfa_mlp_dims *dims) { return native_count(dims->obs_stride) + kTailFloats + 1; }   // + the status entry
static bool env_on(const char *name, bool dflt) {
    const char *e = std::getenv(name);
    return e ? e[0] != '0' : dflt;
}

// 0 = ok, 1 = the grid-wide hand-off of a reduce + Adam launch ran out of its bounded wait (the parameters hold NaN since).  A plain host read.
extern "C" int pfa_ppo_grid_status(void) {
    int *w = grid_status_word();
    return w ? *(volatile int *)w : 0;
}
// Clears the word once it has been reported (the parameters still hold NaN: restore them), so that the recovery the error names —
// PFA_FUSED_ADAM=0 + a checkpoint — can happen inside the same process.  Returns the value it cleared.
extern "C" int pfa_ppo_grid_reset(void) {
    int *w = grid_status_word();
    if (!w) return 0;
    const int was = *(volatile int *)w;
    *(volatile int *)w = 0;
    return was;
}

extern "C" int pfa_ppo_mlp_train_logged(const pfa_experience *exp, int64_t batch_rows, float *params, const pfa_mlp_dims *dims,
                                        const pfa_ppo_hparams *hp, const double *adv_stats, float *grads, float *exp_avg,
                                        float *exp_avg_sq, int64_t opt_step, float lr, float beta1, float beta2, float eps,
                                        float max_grad_norm, int32_t update_epochs, double *losses, void *workspace,
                                        int32_t data_parallel, const double *log_ev4, double *log_out10, int32_t *log_packed,
                                        pfa_stream_t stream);
extern "C" int pfa_ppo_mlp_train(const pfa_experience *exp, int64_t batch_rows, float *params, const pfa_mlp_dims *dims,
                                 const pfa_ppo_hparams *hp, const double *adv_stats, float *grads, float *exp_avg,
                                 float *exp_avg_sq, int64_t opt_step, float lr, float beta1, float beta2, float eps,
                                 float max_grad_norm, int32_t update_epochs, double *losses, void *workspace,
                                 int32_t data_parallel, pfa_stream_t stream) {
    return pfa_ppo_mlp_train_logged(exp, batch_rows, params, dims, hp, adv_stats, grads, exp_avg, exp_avg_sq, opt_step, lr, beta1, beta2, eps,
                                    max_grad_norm, update_epochs, losses, workspace, data_parallel, nullptr, nullptr, nullptr, stream);
}

// pfa_ppo_mlp_train + train()'s report: when the update runs in the one-launch form, its LAST reduce + Adam launch also writes
// log_out10 = { losses[0..5], log_ev4[0..3] } (what pfa_train_log_pack leaves, minus its launch: the report is what the host waits
// for at the end of train()); *log_packed (host) says whether it did — 0: call pfa_train_log_pack as before (two-kernel form, the
// all-reduce as its own step, no optimizer step at all).  log_out10 may be pinned host memory.
extern "C" int pfa_ppo_mlp_train_logged(const pfa_experience *exp, int64_t batch_rows, float *params, const pfa_mlp_dims *dims,
                                        const pfa_ppo_hparams *hp, const double *adv_stats, float *grads, float *exp_avg,
                                        float *exp_avg_sq, int64_t opt_step, float lr, float beta1, float beta2, float eps,
                                        float max_grad_norm, int32_t update_epochs, double *losses, void *workspace,
                                        int32_t data_parallel, const double *log_ev4, double *log_out10, int32_t *log_packed,
                                        pfa_stream_t stream) {
    if (log_packed) *log_packed = 0;
    PFA_REQUIRE(!log_out10 || (log_ev4 && losses), "ppo.train: the report needs the explained-variance sums and the loss accumulator");
    if (int rc = check_update_args(exp, batch_rows, dims, hp)) return rc;
    PFA_REQUIRE(update_epochs >= 0 && opt_step >= 0, "ppo.train: bad epoch / step count");
    PFA_REQUIRE(!data_parallel || dist_ready(), "ppo.train: data_parallel needs pfa_dist_init first");
    const int world = data_parallel ? dist_world() : 1;
    const int64_t mbs = batch_rows / hp->num_minibatches;
    const int64_t global_mbs = mbs * world;
    const int64_t count = mlp_offsets(dims->obs_stride, dims->num_actions).count;
    const double loss_scale = 1.0 / ((double)global_mbs * hp->num_minibatches);
    // PFA_FUSED_ADAM=0: the two-kernel form (sum of the partials, then clip + Adam) for A/B timing; data parallel, the exchange
    // rides inside the fused launch when the peer path's flag-in-data area is open (PFA_FUSED_DP=0: all-reduce as its own step)
    const bool fused_dp_wanted = data_parallel && world > 1 && env_on("PFA_FUSED_DP", true) && p2p_ll_ready(ll_entries_needed(dims));
    const bool fused = env_on("PFA_FUSED_ADAM", true) && reduce_adam_coresident(dims, fused_dp_wanted);
    const bool fused_dp = fused && fused_dp_wanted;
    const bool one_launch = fused && (!data_parallel || fused_dp);   // (a 1-rank communicator keeps exercising the all-reduce call)
    for (int e = 0; e < update_epochs; ++e)
        for (int mb = 0; mb < hp->num_minibatches; ++mb) {
            if (one_launch) {
                int grid = 0;
                if (int rc = launch_grad(exp, batch_rows, mb, params, dims, hp, adv_stats, global_mbs, grads, workspace, stream, &grid)) return rc;
                ++opt_step;
                LlArgs ll{};
                if (fused_dp) ll = p2p_ll_next();
                const bool last = log_out10 && e == update_epochs - 1 && mb == hp->num_minibatches - 1;
                if (int rc = launch_reduce_adam(dims, hp->num_minibatches, grid, params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps,
                                                opt_step, max_grad_norm, losses, loss_scale, workspace, fused_dp ? &ll : nullptr,
                                                (hipStream_t)stream, last ? log_ev4 : nullptr, last ? log_out10 : nullptr))
                    return rc;
                if (last && log_packed) *log_packed = 1;
                continue;
            }
            if (int rc = pfa_ppo_mlp_grad(exp, batch_rows, mb, params, dims, hp, adv_stats, global_mbs, grads, workspace, stream))
                return rc;
            ++opt_step;
            if (data_parallel) {
                // one flat bucket per optimizer step: gradient (already / global rows) + the loss-sum pairs, on this stream
                if (int rc = dist_all_reduce(grads, (size_t)count + kTailFloats, false, (hipStream_t)stream)) return rc;
                if (int rc = pfa_adam_clip_step(params, grads, exp_avg, exp_avg_sq, count, lr, beta1, beta2, eps, opt_step,
                                                max_grad_norm, 1.0f, grads + count, losses, loss_scale, nullptr, 0, stream))
                    return rc;
            } else if (int rc = pfa_adam_clip_step(params, grads, exp_avg, exp_avg_sq, count, lr, beta1, beta2, eps, opt_step,
                                                   max_grad_norm, 1.0f, grads + count, losses, loss_scale,
                                                   norm_partials_of(workspace, dims),
                                                   norm_blocks(dims), stream)) {
                return rc;
            }
        }
    return 0;
}

#ifdef PFA_BF16_TRACE
extern "C" int pfa_probe_bf16_trace(unsigned long long *buf, int tiles) {   // tools/bf16_trace.py (probe builds only)
    PFA_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_bf_trace), &buf, sizeof(buf)));
    PFA_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_bf_trace_tiles), &tiles, sizeof(tiles)));
    return 0;
}
#endif
#ifdef PFA_PROBES
// Probe-only entry (tools/probe_grad.py): same launch as pfa_ppo_mlp_grad for obs_stride 64 with an ablation mask.
extern "C" int pfa_probe_set_trace(unsigned long long *buf, int tiles) {
    PFA_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &buf, sizeof(buf)));
    PFA_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_tiles), &tiles, sizeof(tiles)));
    return 0;
}
extern "C" int pfa_probe_grad(const pfa_experience *exp, int64_t batch_rows, int32_t mb, const float *params,
                              const pfa_mlp_dims *dims, const pfa_ppo_hparams *hp, const double *adv_stats, float *grads,
                              void *workspace, int32_t abl, pfa_stream_t stream) {
    const int64_t mbs = batch_rows / hp->num_minibatches;
    const int grid = grad_grid(mbs, 64);
    RowMap map{mb, hp->num_minibatches, hp->bptt_horizon};
    float *partials = (float *)workspace;
    constexpr size_t lds_bytes = (size_t)GradLds<64, grad_glds(64, 3)>::kFloats * sizeof(float);
#define PFA_PROBE_CASE(A)                                                                                              \
    case A:                                                                                                            \
        PFA_CHECK_HIP(hipFuncSetAttribute((const void *)ppo_mlp_grad_kernel<64, A, 13, false, 3, true>,                                \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));                \
        hipLaunchKernelGGL((ppo_mlp_grad_kernel<64, A, 13, false, 3, true>), dim3(grid), dim3(grad_threads(64)), lds_bytes, (hipStream_t)stream, \
                           *exp, map, (long long)mbs, params, dims->num_actions, 0u, *hp, adv_stats, (double)mbs, partials); \
        break;
    switch (abl) {
        PFA_PROBE_CASE(0) PFA_PROBE_CASE(1) PFA_PROBE_CASE(2) PFA_PROBE_CASE(4) PFA_PROBE_CASE(6) PFA_PROBE_CASE(7)
        default: return -2;
    }
#undef PFA_PROBE_CASE
    PFA_LAUNCH_CHECK();
    (void)grads;
    return 0;
}
#endif

extern "C" int pfa_train_log_sums(const pfa_experience *exp, int64_t batch_rows, int32_t num_envs, const double *losses,
                                  double *out10, void *workspace, pfa_stream_t stream) {
    PFA_REQUIRE(exp && exp->advantages && exp->values && out10 && workspace, "train_log_sums: null buffer");
    PFA_REQUIRE(num_envs >= 1 && batch_rows % num_envs == 0, "train_log_sums: batch must be whole rollout steps");
    double *partial = (double *)workspace;
    hipLaunchKernelGGL(ev_partial_kernel, dim3(kEvBlocks), dim3(256), 0, (hipStream_t)stream, exp->advantages, exp->values,
                       (long long)batch_rows, (int)num_envs, (int)(batch_rows / num_envs), partial);
    PFA_LAUNCH_CHECK();
    hipLaunchKernelGGL(ev_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, losses, out10);
    PFA_LAUNCH_CHECK();
    return 0;
}

// The two halves of pfa_train_log_sums for the data-parallel update, where the four explained-variance sums are known as soon as
// GAE has run and ride the all-reduce of the advantage sums instead of one of their own at the end:
//   pfa_train_ev_sums    out4 = sum y_true, sum y_true^2, sum adv, sum adv^2 of this rank's batch
//   pfa_train_log_pack   out10 = { losses[0..5], ev4[0..3] }  (ev4 = the all-reduced sums)
__global__ void ev_final4_kernel(const double *partial, double *out4) {
    const int q = threadIdx.x;
    if (q < 4) {
        double s = 0.0;
        for (int b = 0; b < kEvBlocks; ++b) s += partial[b * 4 + q];
        out4[q] = s;
    }
}
__global__ void log_pack_kernel(const double *losses, const double *ev4, double *out10) {
    const int q = threadIdx.x;
    if (q < 6) out10[q] = losses ? losses[q] : 0.0;
    else if (q < 10) out10[q] = ev4[q - 6];
}
extern "C" int pfa_train_ev_sums(const pfa_experience *exp, int64_t batch_rows, int32_t num_envs, double *out4, void *workspace,
                                 pfa_stream_t stream) {
    PFA_REQUIRE(exp && exp->advantages && exp->values && out4 && workspace, "train_ev_sums: null buffer");
    PFA_REQUIRE(num_envs >= 1 && batch_rows % num_envs == 0, "train_ev_sums: batch must be whole rollout steps");
    double *partial = (double *)workspace;
    hipLaunchKernelGGL(ev_partial_kernel, dim3(kEvBlocks), dim3(256), 0, (hipStream_t)stream, exp->advantages, exp->values,
                       (long long)batch_rows, (int)num_envs, (int)(batch_rows / num_envs), partial);
    PFA_LAUNCH_CHECK();
    hipLaunchKernelGGL(ev_final4_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, out4);
    PFA_LAUNCH_CHECK();
    return 0;
}
extern "C" int pfa_train_log_pack(const double *losses, const double *ev4, double *out10, pfa_stream_t stream) {
    PFA_REQUIRE(ev4 && out10, "train_log_pack: null buffer");
    hipLaunchKernelGGL(log_pack_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, losses, ev4, out10);
    PFA_LAUNCH_CHECK();
    return 0;
}
