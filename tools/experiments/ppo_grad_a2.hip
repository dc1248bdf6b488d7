// tools/experiments/ppo_grad_a2.hip — ARCHIVED round-3 experiment, not product code and not compiled into the library.
//
// "Kernel A2": the fused PPO gradient kernel re-cut at the loss — the producer wave does forward + heads (+ the dW2v of two
// tiles ago), the consumer wave does the loss, dh, relu' and dW1; hand-off through double-buffered LDS tiles, ONE workgroup
// barrier per tile, the bias folded into a spare observation column, the W1 fragment table unpadded and XOR-swizzled, the
// heads' fragments in registers.  Built into the library (it slots into csrc/ppo_update.hip after kernel A, dispatch macro at
// the end of this file), parity-green against the oracle (tests/test_gpu_ppo.py, test_gpu_parity_full.py: 43 passed, weights
// within 1.5e-8) and timed with tools/variant_bench.py on the bench shape:
//       kernel A  (round 2 order)           61.4 us / launch
//       kernel A  + pipelined fragment loads 59.1 us            <- what the product runs
//       kernel A2                            60.8 us
// i.e. no gain from removing the second barrier and from putting the loss under the other wave's MFMAs: the launch is bound by
// the instruction issue of a SIMD's two waves (MFMA + VALU cycles add, profiles/r02_ubench_simd_share.txt), not by the hand-off.
// Kept for the record; DESIGN.md section 3.4 has the reasoning.
// ------------------------------------------------------------------------------------------------------------------------------
// Kernel A2 (round 3): the same products, re-cut so that the MFMA pipe of a SIMD never sits idle behind the loss.
//
// What round 2's timeline showed (profiles/r02_grad_timeline_pairs.txt): producer and consumer of a pair start their shares at
// the same barrier; the consumer (128 MFMAs, dense) is through before the producer reaches the per-row PPO loss, so the loss —
// ~240 dependent VALU instructions, latency-bound, 1.3 us per tile — ran alone on the SIMD, and so did the publish window between
// the two barriers of a tile (0.5 us).  Together a quarter of the tile time with the matrix pipe idle.
//
// A2 cuts the tile at the loss instead of after it, hands over through DOUBLE-BUFFERED LDS tiles, and needs ONE barrier per tile:
//   producer  iteration i :  dW2v(i-2) from hidden(i-2), dout(i-2)      32 MFMA   (its operands are two barriers old)
//                            stage X(i), hidden^T = W1 X^T (+ bias col) KKU x 8
//                            ReLU, out^T = W2v hidden^T                  32
//                            publish hidden(i) [row][u], out^T(i) fragment -> buffers i & 1              | barrier B_i
//   consumer  after B_i   :  loss(i) from out^T(i) and its own prefetched row scalars  (VALU, runs under the producer's MFMAs)
//                            dout(i) -> LDS [row][o] (for the producer's dW2v two iterations later)
//                            dh = dout W2v (32), relu', db1, column; dW1^T += X^T dh (KTM x 32)          | barrier B_{i+1}
// so in every window the SIMD has 168 producer MFMAs and 128 consumer MFMAs to interleave with the loss's dependency stalls, and
// nobody waits for a publish.  Hazards: a buffer of parity b written in iteration i is read by the consumer before B_{i+1} and by
// the producer's dW2v in iteration i+2 before it publishes hidden(i+2) into the same buffer (program order); dout(i) is written
// before B_{i+1}, read in iteration i+2, overwritten after B_{i+2}.
//
// LDS (obs rows <= 64 floats, 159 KB): per pair X ring 2 x 16 x (DP+2), hidden 2 x 16 x 132, out^T 2 x 256, dout 2 x 16 x 20;
// shared: the W1 fragment table WITHOUT padding — lane stride DP/4 floats, the lane's b128 groups XOR-swizzled by
// (lane / (16/NG)) % NG so that 16 lanes still cover all 64 banks — and the dh B-fragment table.  The heads' A fragments live in
// producer registers (constant per launch).  The bias needs no table: observation rows narrower than their stride have a spare
// column; staging writes 1.0 there and the W1 table carries b1 in that column, so hidden^T starts from zero accumulators.
// (Rows that fill their stride keep kernel A.)  The MFMA count, every accumulation order within an accumulator and the loss code
// are kernel A's: gradients are bit-identical to it (tools/variant_bench.py compares).
template <int DP>
struct Grad2Lds {
    static constexpr int XS = XTile<DP>::XS;
    static constexpr int KS = DP / 4;          // W1 fragment floats per lane
    static constexpr int NG = KS / 4;          // b128 groups per lane: 1, 2, 4
    static constexpr int HS = kHidden + 4;
    static constexpr int DS = 20;
    static constexpr int kXT = 0;                          // + slot * 16 * XS
    static constexpr int kHT = 2 * 16 * XS;                // + buf * 16 * HS     hidden [row][u] (post-relu)
    static constexpr int kOT = kHT + 2 * 16 * HS;          // + buf * 256         out^T C fragment, lane-linear
    static constexpr int kDT = kOT + 2 * 256;              // + buf * 16 * DS     dout [row][o]
    static constexpr int kPairFloats = kDT + 2 * 16 * DS;
    static constexpr int kW1Floats = kMT * 64 * KS;
    static constexpr int kTabFloats = kMT * 64 * 4;
    static constexpr int kFloats = 4 * kPairFloats + kW1Floats + kTabFloats;
};

template <int DP, int KKU, bool MH, int KTM, bool COL>
__global__ void __launch_bounds__(512, 2)
    ppo_mlp_grad2_kernel(pfa_experience ex, RowMap map, long long mb_rows, const float *params, int a, uint32_t heads, int obs_dim,
                         pfa_ppo_hparams hp, const double *adv_stats /* [nmb][2] */, double global_rows, float *partials) {
    using L = Grad2Lds<DP>;
    using NL = NativeLayout<DP>;
    constexpr int XS = L::XS, HS = L::HS, DS = L::DS, KT = DP / 16, KS = L::KS, NG = L::NG, V = DP / 4;
    constexpr int NLD = (16 * V + 63) / 64;
    constexpr int kPairs = 4, kThreads = 512;
    static_assert(DP <= 64 && (NG == 1 || NG == 2 || NG == 4), "kernel A2 is the narrow-row form");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = lane_id(), wv = wave_id(), c = lane & 15, g = lane >> 4;
    const int pair = wv & (kPairs - 1);
    const bool producer = wv < kPairs;
    float *pl = lds + pair * L::kPairFloats;
    float *w1t = lds + kPairs * L::kPairFloats;
    float *w2bt = w1t + L::kW1Floats;
    const MlpOffsets off = mlp_offsets(DP, a);

    // Fragment tables: W1 with the bias in column obs_dim, group-swizzled; the B fragments of dh = dout . W2v.
    {
        constexpr int N1 = kMT * 64 * KS / kThreads, N2 = kMT * 64 * 4 / kThreads;
        float tv[N2], t1[N1];
#pragma unroll
        for (int j = 0; j < N2; ++j) {
            const int i = threadIdx.x + j * kThreads;
            const int r = i & 3, ln = (i >> 2) & 63, m = i >> 8, cc = ln & 15, gg = ln >> 4;
            tv[j] = w2v_at(params, off, a, 4 * gg + r, 16 * m + cc);    // B[k-slot gg][j=u=16m+cc] for o = 4gg+r
        }
#pragma unroll
        for (int j = 0; j < N1; ++j) {
            const int i = threadIdx.x + j * kThreads;
            const int kk = i % KS, ln = (i / KS) & 63, m = i / (KS * 64), cc = ln & 15, gg = ln >> 4;
            const int col = 4 * kk + gg;
            t1[j] = col == obs_dim ? params[off.b1 + 16 * m + cc] : params[off.w1 + (16 * m + cc) * DP + col];
        }
#pragma unroll
        for (int j = 0; j < N1; ++j) {
            const int i = threadIdx.x + j * kThreads;
            const int kk = i % KS, ln = (i / KS) & 63, m = i / (KS * 64);
            const int sw = (ln / (16 / NG)) % NG;
            w1t[(m * 64 + ln) * KS + 4 * ((kk >> 2) ^ sw) + (kk & 3)] = t1[j];
        }
#pragma unroll
        for (int j = 0; j < N2; ++j) w2bt[threadIdx.x + j * kThreads] = tv[j];
    }

    const long long tiles = mb_rows / 16;
    const long long pair_global = (long long)blockIdx.x * kPairs + pair;
    const long long pair_count = (long long)gridDim.x * kPairs;
    const int J = (int)((tiles + pair_count - 1) / pair_count);  // same for every pair: all waves run the same barriers
    const bool aligned = (map.horizon & 15) == 0;
    float *red = lds + (pair & 1) * NL::kCount;  // epilogue reduction: buffer 0 even pairs, buffer 1 odd pairs

    if (producer) {
        // ------------------------------------------------------------------------------------------ producer
        f32x4 w4h[kMT];   // A fragments of the heads, constant per launch: W2v[o = c][u = 16m + 4g + r]
        float bo[4];
#pragma unroll
        for (int m = 0; m < kMT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) w4h[m][r] = w2v_at(params, off, a, c, 16 * m + 4 * g + r);
#pragma unroll
        for (int r = 0; r < 4; ++r) bo[r] = b2v_at(params, off, a, 4 * g + r);
        f32x4 acc_dw2[kMT];
#pragma unroll
        for (int m = 0; m < kMT; ++m) acc_dw2[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int sw = (lane / (16 / NG)) % NG;

        float4 xpre[NLD];  // register prefetch of the next tile's X rows, issued a whole tile ahead
        auto prefetch = [&](long long tile) {
            const bool ok = tile < tiles;
            const unsigned first = ok ? map.tile_first((unsigned)tile) : 0u;
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
        };
        // dW2v^T[u][o] += hidden^T . dout of tile jj, both operands from the hand-off buffers in A/B fragment order
        auto dw2_tile = [&](int jj) {
            if (jj < 0 || pair_global + (long long)jj * pair_count >= tiles) return;
            const float *hsB = pl + L::kHT + (jj & 1) * 16 * HS, *dtB = pl + L::kDT + (jj & 1) * 16 * DS;
            float dfrag[4], hrow[kMT][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) dfrag[r] = dtB[(4 * g + r) * DS + c];              // B: k-slot g <-> row 4g+r, j = o = c
#pragma unroll
            for (int m = 0; m < kMT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) hrow[m][r] = hsB[(4 * g + r) * HS + 16 * m + c];  // A: hidden[row=4g+r][u=16m+c]
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m = 0; m < kMT; ++m) acc_dw2[m] = mfma16(hrow[m][r], dfrag[r], acc_dw2[m]);
        };
        prefetch(pair_global);
        __syncthreads();  // fragment tables ready

        for (int j = 0; j < J; ++j) {
            float *xs = pl + L::kXT + (j & 1) * 16 * XS;
            float *hsB = pl + L::kHT + (j & 1) * 16 * HS, *otB = pl + L::kOT + (j & 1) * 256;
            // ---- stage X(j) with the bias column, then the lagging dW2v while the stores drain -------------------------------
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
            prefetch(pair_global + (long long)(j + 1) * pair_count);  // lands during this tile
            wave_lds_fence();
            if (lane < 16) xs[lane * XS + obs_dim] = 1.0f;           // the bias column (after the row stores: LDS is in order per wave)
            wave_lds_fence();
            dw2_tile(j - 2);

            // ---- hidden^T = W1 X^T, software-pipelined over the k-groups (see kernel A) -----------------------------------
            constexpr int NK4 = (KKU + 3) / 4;
            f32x4 h[kMT], wq[2][kMT];
            float xb[2][4];
            auto load_group = [&](int k4, int b) {
#pragma unroll
                for (int q = 0; q < 4; ++q) xb[b][q] = xs[c * XS + 4 * (4 * k4 + q) + g];
#pragma unroll
                for (int m = 0; m < kMT; ++m) wq[b][m] = *reinterpret_cast<const f32x4 *>(w1t + (m * 64 + lane) * KS + 4 * (k4 ^ sw));
            };
            load_group(0, 0);
#pragma unroll
            for (int m = 0; m < kMT; ++m) h[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k4 = 0; k4 < NK4; ++k4) {
                if (k4 + 1 < NK4) load_group(k4 + 1, (k4 + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int m = 0; m < kMT; ++m)
                        if (4 * k4 + q < KKU) h[m] = mfma16(wq[k4 & 1][m][q], xb[k4 & 1][q], h[m]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < kMT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) h[m][r] = fmaxf(h[m][r], 0.0f);
            f32x4 out;
            {
                f32x4 o[4] = {f32x4{bo[0], bo[1], bo[2], bo[3]}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f},
                              f32x4{0.f, 0.f, 0.f, 0.f}};  // four independent chains, kernel A's order
#pragma unroll
                for (int m0 = 0; m0 < kMT; m0 += 4)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int q = 0; q < 4; ++q) o[q] = mfma16(w4h[m0 + q][r], h[m0 + q][r], o[q]);
                out = (o[0] + o[1]) + (o[2] + o[3]);  // out^T[o = 4g + r][row = c]
            }
            // ---- publish hidden(j) [row][u] and the out^T fragment ---------------------------------------------------------
#pragma unroll
            for (int m = 0; m < kMT; ++m) *reinterpret_cast<f32x4 *>(hsB + c * HS + 16 * m + 4 * g) = h[m];
            *reinterpret_cast<f32x4 *>(otB + lane * 4) = out;
            __syncthreads();  // B_j
        }
        dw2_tile(J - 2);   // its dout was complete at B_{J-1}
        __syncthreads();   // B_J: the consumer is through tile J-1
        dw2_tile(J - 1);
        __syncthreads();   // every wave has stopped reading the hand-off buffers: the reduction may overwrite them

        // ---- epilogue: producers own dW2v -----------------------------------------------------------------------------------
        for (int turn = 0; turn < kPairs / 2; ++turn) {
            if ((pair >> 1) == turn) {
                const bool first = turn == 0;
#pragma unroll
                for (int m = 0; m < kMT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = NL::kDw2 + (m * 4 + r) * 64 + lane;
                        red[idx] = (first ? 0.0f : red[idx]) + acc_dw2[m][r];
                    }
            }
            __syncthreads();
        }
    } else {
        // ------------------------------------------------------------------------------------------ consumer
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

        f32x4 acc_dw1[KTM][kMT];
        float acc_col[kMT], db1[kMT], db2[4], stats[6];
#pragma unroll
        for (int m = 0; m < kMT; ++m) {
#pragma unroll
            for (int kt = 0; kt < KTM; ++kt) acc_dw1[kt][m] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc_col[m] = db1[m] = 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) db2[r] = 0.0f;
#pragma unroll
        for (int i = 0; i < 6; ++i) stats[i] = 0.0f;

        RowScalars rspre;   // per-row scalars (lane c <-> row c, replicated over the 4 lane groups), a tile ahead
        auto prefetch = [&](long long tile) {
            rspre = RowScalars{0, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (tile < tiles) {
                const unsigned first = map.tile_first((unsigned)tile);
                const unsigned fr = map.tile_row((unsigned)tile, first, c, aligned);
                rspre = RowScalars{ex.actions[fr], ex.logprobs[fr], ex.values[fr], ex.advantages[fr], ex.returns[fr], 1.0f};
            }
        };
        auto process = [&](int jj, const RowScalars &rs) {
            if (pair_global + (long long)jj * pair_count >= tiles) return;
            const float *xs = pl + L::kXT + (jj & 1) * 16 * XS, *hsB = pl + L::kHT + (jj & 1) * 16 * HS;
            const float *otB = pl + L::kOT + (jj & 1) * 256;
            float *dtB = pl + L::kDT + (jj & 1) * 16 * DS;
            // the LDS reads are issued ahead of their use: `out`, the X fragments and the first half of the hidden tile / W2v
            // fragments before the loss (only `out` is needed by it, the rest lands under it), the second half under the first
            // half's MFMAs.  Two halves of four hidden tiles keep the consumer inside its 256 registers.
            const f32x4 out = *reinterpret_cast<const f32x4 *>(otB + lane * 4);
            float xa[KTM][4], xc[4], hrow[2][4][4];
            f32x4 wb[2][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int kt = 0; kt < KTM; ++kt) xa[kt][r] = xs[(4 * g + r) * XS + 16 * kt + c];
                xc[r] = COL ? xs[(4 * g + r) * XS + 16 * KTM] : 0.0f;
            }
            auto load_half = [&](int hf) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    wb[hf][q] = *reinterpret_cast<const f32x4 *>(w2bt + ((4 * hf + q) * 64 + lane) * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) hrow[hf][q][r] = hsB[(4 * g + r) * HS + 16 * (4 * hf + q) + c];  // hidden[row=4g+r][u=16m+c]
                }
            };
            load_half(0);
            __builtin_amdgcn_sched_barrier(0);
            const LossOut lo = ppo_loss_tile<MH, !MH>(out, rs, a, heads, g, hp, adv_mean, adv_rden, inv_rows);
            const f32x4 dout = lo.dout;
            *reinterpret_cast<f32x4 *>(dtB + c * DS + 4 * g) = dout;  // dout[row = c][o = 4g..4g+3] for the producer's dW2v
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
            // dh[row][u] = dout . W2v (A = dout^T's C fragment, B = w2bt), relu' from the hidden tile, db1, the column,
            // dW1^T[k][u] += X^T . dh (A = X tile: i = k = 16kt + c, k-slot g <-> row 4g+r;  B = dh's own C fragment)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                f32x4 dh[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) dh[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (hf == 0) load_half(1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q) dh[q] = mfma16(dout[r], wb[hf][q][r], dh[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dh[q][r] = hrow[hf][q][r] > 0.0f ? dh[q][r] : 0.0f;  // relu'
                        db1[4 * hf + q] += dh[q][r];
                        if (COL) acc_col[4 * hf + q] = fmaf(xc[r], dh[q][r], acc_col[4 * hf + q]);
                    }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int kt = 0; kt < KTM; ++kt) acc_dw1[kt][4 * hf + q] = mfma16(xa[kt][r], dh[q][r], acc_dw1[kt][4 * hf + q]);
            }
        };
        prefetch(pair_global);
        __syncthreads();  // fragment tables ready (same barrier as the producers')
        for (int j = 0; j < J; ++j) {
            __syncthreads();  // B_j: hidden(j), out^T(j) published
            const RowScalars rs = rspre;
            prefetch(pair_global + (long long)(j + 1) * pair_count);
            process(j, rs);
        }
        __syncthreads();  // B_J
        __syncthreads();  // the producers' last dW2v has read its operands

        // ---- epilogue: consumers own dW1, db1, db2v, stats -----------------------------------------------------------------
#pragma unroll
        for (int m = 0; m < kMT; ++m) {
            db1[m] += __shfl_xor(db1[m], 16, 64);
            db1[m] += __shfl_xor(db1[m], 32, 64);
            if (COL) acc_col[m] = gsum<true>(acc_col[m]);   // over the lane groups: all 16 rows of the tile
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int sft = 1; sft < 16; sft <<= 1) db2[r] += __shfl_xor(db2[r], sft, 64);
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int sft = 1; sft < 16; sft <<= 1) stats[i] += __shfl_xor(stats[i], sft, 64);
        for (int turn = 0; turn < kPairs / 2; ++turn) {
            if ((pair >> 1) == turn) {
                const bool first = turn == 0;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int m = 0; m < kMT; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int idx = NL::kDw1 + ((kt * kMT + m) * 4 + r) * 64 + lane;
                            float val = 0.0f;   // k-tiles beyond KTM: the trailing column in slot (kt = KTM, r = 0, g = 0), else padding
                            if (kt < KTM) val = acc_dw1[kt < KTM ? kt : 0][m][r];
                            else if (COL && kt == KTM && r == 0 && g == 0) val = acc_col[m];
                            red[idx] = (first ? 0.0f : red[idx]) + val;
                        }
                if (g == 0) {
#pragma unroll
                    for (int m = 0; m < kMT; ++m) {
                        const int idx = NL::kDb1 + 16 * m + c;
                        red[idx] = (first ? 0.0f : red[idx]) + db1[m];
                    }
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
    }
    float *dst = partials + (size_t)blockIdx.x * NL::kCount;
    for (int i = threadIdx.x; i < NL::kCount; i += kThreads) dst[i] = lds[i] + lds[NL::kCount + i];
}


// ---- dispatch that went with it (inside pfa_ppo_mlp_grad) --------------------------------------------------------------------
// #define PFA_LAUNCH_GRAD2(DPV, KKUV, MHV, KTMV, COLV) { lds_bytes = Grad2Lds<DPV>::kFloats * 4; hipFuncSetAttribute(...MaxDynamicSharedMemorySize...);
//     hipLaunchKernelGGL((ppo_mlp_grad2_kernel<DPV, KKUV, MHV, KTMV, COLV>), dim3(grid), dim3(512), lds_bytes, stream, *exp, map, mbs, params,
//                        dims->num_actions, dims->heads, dims->obs_dim, *hp, adv_stats, (double)global_mb_rows, partials); }
// used for obs_dim < obs_stride and (obs_stride <= 32, or obs_stride == 64 with obs_dim == 49, one Discrete head: <64, 13, false, 3, true>)
