// lstm_seq.hip — the recurrent policy in TRAINING mode (clean_pufferl.py:186-193 -> models.py:86-111 with
// x of shape [B, TT, ...]): forward through the bptt_horizon steps of a minibatch and back-propagation through time,
// one persistent workgroup per 32 rows (two 16-row MFMA tiles) for all steps — rows (= bptt segments) are independent.
//
//   lstm_seq_fwd_kernel   per step: encode_observations (kept as xe) -> gate product against the packed [W_ih | W_hh]
//                         streamed from L2 (lstm_tile.hpp) -> cell update; keeps what autograd would keep:
//                         gate activations (i,f,g,o), c_t, h_t.  Time-major buffers [t][R][...].
//   lstm_pack_bwd_kernel  Wcat^T in A-fragment order for the backward product
//   lstm_seq_bwd_kernel   per step, t = Th-1 .. 0: cell backward (d loss/d h_t = heads path + recurrent path) ->
//                         dG_t [rows][512] (kept for the weight-gradient contractions, csrc/gemm.hip) ->
//                         [dxe_t | dh_{t-1}]^T = Wcat^T dG_t^T on MFMA; relu' applied to dxe; dh_{t-1}, dc stay in
//                         registers.  Also accumulates the bias gradients (column sums of dG and dxe) per workgroup.
// Roofline: fp32 MFMA, 2*256*512 flop per row and step in each direction (+ 2*64*128 for the encoder forward).
#include "common.hpp"
#include "lane_ops.hpp"
#include "lstm_tile.hpp"
#include "mlp_tile.hpp"

namespace pfa {

constexpr int kSeqRT = 2;                 // row tiles per workgroup
constexpr int kSeqRows = 16 * kSeqRT;
constexpr int kDGS = kLG + 4;             // LDS row stride of a dG tile (floats)
constexpr int kDGTile = 16 * kDGS;
constexpr int kBiasCols = kLG + kLH;      // per-workgroup partial column sums: dG (512) then dxe (128)

template <int DP>
struct SeqFwdLds {
    float xs[kSeqRT][XTile<DP>::kFloats];
    float xh[2][kSeqRT][kXHTile];
    float gbias[kLG];
};

template <int DP>
__global__ void __launch_bounds__(kLstmThreads) lstm_seq_fwd_kernel(const float *__restrict__ obs_tm, long long R, int Th,
                                                                   const float *__restrict__ params, int a,
                                                                   const float4 *__restrict__ wpack, float *__restrict__ xe,
                                                                   float *__restrict__ gates_act, float *__restrict__ Hs,
                                                                   float *__restrict__ Cs, int init_slot) {
    constexpr int V = DP / 4;
    __shared__ SeqFwdLds<DP> L;
    LstmFrags<DP> w;
    w.load(params, a);
    stage_gate_bias(params, DP, a, L.gbias);
    const int wv = wave_id(), c = lane_id() & 15, g = lane_id() >> 4;
    const float4 *wp = wpack + (size_t)__builtin_amdgcn_readfirstlane(wv) * 16 * 8 * 64;
    const long long first = (long long)blockIdx.x * kSeqRows;
    const int uoff = 32 * wv + 4 * g;
    f32x4 cst[kSeqRT][2];
    // initial state: slot `init_slot` of Hs / Cs (0: it is already where the backward pass and dW_hh read it; k > 0: the previous
    // minibatch's final state, clean_pufferl.py:188-191; < 0: lstm_state = None, :176), copied into slot 0 by the workgroup that owns
    // the rows — no copy / fill launch in front of this one
    const float *h0 = init_slot < 0 ? nullptr : Hs + (size_t)init_slot * R * kLH;
    const float *c0 = init_slot < 0 ? nullptr : Cs + (size_t)init_slot * R * kLH;
#pragma unroll
    for (int rt = 0; rt < kSeqRT; ++rt) {
        load_hstate(h0, first + 16 * rt, R, L.xh[0][rt]);
        load_cstate(c0, first + 16 * rt + c, c0 && first + 16 * rt + c < R, cst[rt]);
        if (init_slot != 0) {
            store_hstate(Hs, first + 16 * rt, R, L.xh[0][rt]);    // (the same thread wrote these LDS elements)
            store_cstate(Cs, first + 16 * rt + c, first + 16 * rt + c < R, cst[rt]);
        }
        lstm_stage_obs<DP>(obs_tm, first + 16 * rt, R, L.xs[rt]);
    }
    __syncthreads();

    // What step t keeps for the backward pass leaves the registers during step t+1's gate product, two stores per
    // weight group (stream_product's side slot): per row tile 8 gate-activation float4, 2 of h_t, 2 of c_t.
    constexpr int kPend = 12;
    float4 pend[kSeqRT][kPend];
    auto flush = [&](int k, int tp) {  // store #k (0 .. 2*kPend-1) of step tp
        const int rt = k / kPend, j = k % kPend;
        const long long row = first + 16 * rt + c;
        if (row >= R) return;
        if (j < 8)
            *reinterpret_cast<float4 *>(gates_act + ((size_t)tp * R + row) * kLG + 128 * (j >> 1) + 16 * (j & 1) + uoff) = pend[rt][j];
        else if (j < 10)
            *reinterpret_cast<float4 *>(Hs + ((size_t)(tp + 1) * R + row) * kLH + 16 * (j - 8) + uoff) = pend[rt][j];
        else
            *reinterpret_cast<float4 *>(Cs + ((size_t)(tp + 1) * R + row) * kLH + 16 * (j - 10) + uoff) = pend[rt][j];
    };

    for (int t = 0; t < Th; ++t) {
        const int cur = t & 1;
        float4 ring[kGateRing][8];
        const Stream ws = stream_begin(wp);
        ring_prefetch<8, kGateRing>(ws, ring);
#pragma unroll
        for (int rt = 0; rt < kSeqRT; ++rt) {
            const long long row = first + 16 * rt + c;
            lstm_encode<DP>(w, L.xs[rt], L.xh[cur][rt], row < R ? xe + ((size_t)t * R + row) * kLH : nullptr);
        }
        __syncthreads();
        // next step's observation rows: global -> registers now, registers -> LDS after the gate product
        constexpr int NX = (kSeqRows * V + kLstmThreads - 1) / kLstmThreads;
        float4 nx[NX];
        if (t + 1 < Th) {
#pragma unroll
            for (int q = 0; q < NX; ++q) {
                const int idx = threadIdx.x + q * kLstmThreads;
                const long long row = first + idx / V;
                nx[q] = (idx < kSeqRows * V && row < R)
                            ? *reinterpret_cast<const float4 *>(obs_tm + ((size_t)(t + 1) * R + row) * DP + 4 * (idx % V))
                            : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        f32x4 acc[kSeqRT][8];
#pragma unroll
        for (int rt = 0; rt < kSeqRT; ++rt)
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool have_pend = t > 0;
        stream_product<16, 8, kSeqRT, kXHTile, kGateRing>(ws, L.xh[cur][0] + c * kXHS + 4 * g, ring, acc, [&](int gq) {
            if (have_pend && 2 * gq + 1 < kSeqRT * kPend) {
                flush(2 * gq, t - 1);
                flush(2 * gq + 1, t - 1);
            }
        });
#pragma unroll
        for (int rt = 0; rt < kSeqRT; ++rt) {
            f32x4 hn[2];
            lstm_cell(acc[rt], L.gbias, cst[rt], hn);
            float4 *dst = reinterpret_cast<float4 *>(L.xh[cur ^ 1][rt] + c * kXHS + kLH + uoff);
            dst[0] = make_float4(hn[0][0], hn[0][1], hn[0][2], hn[0][3]);
            dst[4] = make_float4(hn[1][0], hn[1][1], hn[1][2], hn[1][3]);
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) pend[rt][ct] = make_float4(acc[rt][ct][0], acc[rt][ct][1], acc[rt][ct][2], acc[rt][ct][3]);
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                pend[rt][8 + sub] = make_float4(hn[sub][0], hn[sub][1], hn[sub][2], hn[sub][3]);
                pend[rt][10 + sub] = make_float4(cst[rt][sub][0], cst[rt][sub][1], cst[rt][sub][2], cst[rt][sub][3]);
            }
        }
        if (t + 1 < Th) {
            constexpr int XS = XTile<DP>::XS;
#pragma unroll
            for (int q = 0; q < NX; ++q) {
                const int idx = threadIdx.x + q * kLstmThreads;
                if (idx >= kSeqRows * V) continue;
                const int r = idx / V, c4 = idx % V;
                float2 *d = reinterpret_cast<float2 *>(&L.xs[r >> 4][(r & 15) * XS + 4 * c4]);
                d[0] = make_float2(nx[q].x, nx[q].y);
                d[1] = make_float2(nx[q].z, nx[q].w);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < kSeqRT * kPend; ++k) flush(k, Th - 1);
}

// dst float4 index ((w*32 + cq)*4 + kt)*64 + lane  <-  { Wcat[16cq + 4g + s][k(w, kt, c)] : s = 0..3 },
// k(w, kt, i) = 128*(kt >> 1) + 32w + 16*(kt & 1) + i   (kt 0,1: the xe half -> dxe;  kt 2,3: the h half -> dh_{t-1})
__global__ void __launch_bounds__(256) lstm_pack_bwd_kernel(const float *__restrict__ params, int dp, int a, float4 *__restrict__ dst,
                                                            float4 *__restrict__ fwd_dst = nullptr) {
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= kGatePackFloats / 4) {   // second half of the grid (pfa_lstm_pack_both): the forward pack of csrc/lstm_fused.hip
        idx -= kGatePackFloats / 4;
        if (fwd_dst && idx < kGatePackFloats / 4) fwd_dst[idx] = lstm_pack_fwd_elem(params, dp, a, idx);
        return;
    }
    const LstmOffsets lo = lstm_offsets(dp, a);
    const int lane = idx & 63, kt = (idx >> 6) & 3, cq = (idx >> 8) & 31, w = idx >> 13;
    const int c = lane & 15, g = lane >> 4;
    const int k = 32 * w + 16 * (kt & 1) + c;
    const float *src = params + ((kt >> 1) ? lo.w_hh : lo.w_ih) + k;
    const int col = 16 * cq + 4 * g;
    dst[idx] = make_float4(src[(col + 0) * kLH], src[(col + 1) * kLH], src[(col + 2) * kLH], src[(col + 3) * kLH]);
}

struct SeqBwdLds {
    float dg[kSeqRT][kDGTile];
};

// FULL: the minibatch is a whole number of 32-row tiles (every BASELINE shape) — no row guards, so every load and store of the loop is
// unconditional and the compiler can COUNT the stores a wait skips (`vmcnt(n)`); with guards it cannot, and each wait for the prefetched
// inputs became a full `vmcnt(0)` that also waits for the 16 dG stores issued just before (tools/isa_vmcnt_scan.py).
template <bool FULL>
__global__ void __launch_bounds__(kLstmThreads) lstm_seq_bwd_kernel(const float *__restrict__ gates_act, const float *__restrict__ Cs,
                                                                   const float *__restrict__ xe, const float *__restrict__ dh_heads,
                                                                   long long R, int Th, const float4 *__restrict__ wpack_bwd,
                                                                   float *__restrict__ dG, float *__restrict__ dxe,
                                                                   float *__restrict__ bias_partial) {
    __shared__ SeqBwdLds L;
    const int lane = lane_id(), wv = wave_id(), c = lane & 15, g = lane >> 4;
    const float4 *wp = wpack_bwd + (size_t)__builtin_amdgcn_readfirstlane(wv) * 32 * 4 * 64;
    const long long first = (long long)blockIdx.x * kSeqRows;
    const int uoff = 32 * wv + 4 * g;  // this lane's units: uoff + 16 sub + r
    constexpr int kBwdRing = 8;        // 4 fragments per group: keep 7 groups (3 us) of the stream in flight

    f32x4 dc[kSeqRT][2], dhrec[kSeqRT][2], xsum[2];
    float bcol0 = 0.0f, bcol1 = 0.0f;  // column sums of dG (gate bias gradient) for columns tid and tid + 256
#pragma unroll
    for (int rt = 0; rt < kSeqRT; ++rt)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) dc[rt][sub] = dhrec[rt][sub] = f32x4{0.f, 0.f, 0.f, 0.f};
    xsum[0] = xsum[1] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Inputs of the cell backward of one step, per row tile: 8 gate activations (q, sub), c_{t-1} (2), dh_heads (2); the
    // encoder output xe (2) masks dxe after the product.  Step t-1's set is fetched during step t's product, one load per
    // weight group (stream_product's side slot); c_t is last step's c_{t-1}.
    constexpr int kIn = 12;
    float4 in[kSeqRT][kIn], xv[kSeqRT][2], ccur[kSeqRT][2];
    auto fetch = [&](int k, int tt) {  // load #k (0 .. 2*kIn-1) of step tt
        const int rt = k / kIn, j = k % kIn;
        const long long row = first + 16 * rt + c;
        if (!FULL && row >= R) {
            in[rt][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            return;
        }
        const size_t tr = (size_t)tt * R + row;
        if (j < 8)
            in[rt][j] = *reinterpret_cast<const float4 *>(gates_act + tr * kLG + 128 * (j >> 1) + 16 * (j & 1) + uoff);
        else if (j < 10)
            in[rt][j] = *reinterpret_cast<const float4 *>(Cs + tr * kLH + 16 * (j - 8) + uoff);  // c_{tt-1}
        else
            in[rt][j] = *reinterpret_cast<const float4 *>(dh_heads + tr * kLH + 16 * (j - 10) + uoff);
    };
    auto fetch_x = [&](int k, int tt) {  // load #k (0..3) of xe at step tt
        const int rt = k >> 1, sub = k & 1;
        const long long row = first + 16 * rt + c;
        xv[rt][sub] = (FULL || row < R) ? *reinterpret_cast<const float4 *>(xe + ((size_t)tt * R + row) * kLH + 16 * sub + uoff) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
#pragma unroll
    for (int k = 0; k < kSeqRT * kIn; ++k) fetch(k, Th - 1);
#pragma unroll
    for (int rt = 0; rt < kSeqRT; ++rt)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const long long row = first + 16 * rt + c;
            ccur[rt][sub] = (FULL || row < R) ? *reinterpret_cast<const float4 *>(Cs + ((size_t)Th * R + row) * kLH + 16 * sub + uoff) : make_float4(0.f, 0.f, 0.f, 0.f);
        }

    for (int t = Th - 1; t >= 0; --t) {
        float4 ring[kBwdRing][4];
        const Stream ws = stream_begin(wp);
        ring_prefetch<4, kBwdRing>(ws, ring);  // ahead of this step's stores in the memory queue
        // ---- cell backward, elementwise on (row c, units uoff + 16 sub + r) ------------------------------------
#pragma unroll
        for (int rt = 0; rt < kSeqRT; ++rt) {
            const long long row = first + 16 * rt + c;
            const bool ok = FULL || row < R;
            float4 *dgo = reinterpret_cast<float4 *>(dG + ((size_t)t * R + (ok ? row : 0)) * kLG + uoff);
            float *dgl = L.dg[rt] + c * kDGS + uoff;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const float4 vi = in[rt][0 + sub], vf = in[rt][2 + sub], vg = in[rt][4 + sub], vo = in[rt][6 + sub];
                const float4 vcp = in[rt][8 + sub], vcn = ccur[rt][sub], vdh = in[rt][10 + sub];
                const float ai[4] = {vi.x, vi.y, vi.z, vi.w}, af[4] = {vf.x, vf.y, vf.z, vf.w};
                const float ag[4] = {vg.x, vg.y, vg.z, vg.w}, ao[4] = {vo.x, vo.y, vo.z, vo.w};
                const float acp[4] = {vcp.x, vcp.y, vcp.z, vcp.w}, acn[4] = {vcn.x, vcn.y, vcn.z, vcn.w};
                const float adh[4] = {vdh.x, vdh.y, vdh.z, vdh.w};
                f32x4 gi, gf, gg, go;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float tc = tanh_f(acn[r]);
                    const float dhx = adh[r] + dhrec[rt][sub][r];  // heads path + recurrent path
                    const float d_o = dhx * tc;
                    const float d_c = dhx * ao[r] * (1.0f - tc * tc) + dc[rt][sub][r];
                    gi[r] = d_c * ag[r] * ai[r] * (1.0f - ai[r]);
                    gf[r] = d_c * acp[r] * af[r] * (1.0f - af[r]);
                    gg[r] = d_c * ai[r] * (1.0f - ag[r] * ag[r]);
                    go[r] = d_o * ao[r] * (1.0f - ao[r]);
                    dc[rt][sub][r] = d_c * af[r];
                }
                const float4 fi = make_float4(gi[0], gi[1], gi[2], gi[3]), ff = make_float4(gf[0], gf[1], gf[2], gf[3]);
                const float4 fg = make_float4(gg[0], gg[1], gg[2], gg[3]), fo = make_float4(go[0], go[1], go[2], go[3]);
                *reinterpret_cast<float4 *>(dgl + 0 + 16 * sub) = fi;
                *reinterpret_cast<float4 *>(dgl + 128 + 16 * sub) = ff;
                *reinterpret_cast<float4 *>(dgl + 256 + 16 * sub) = fg;
                *reinterpret_cast<float4 *>(dgl + 384 + 16 * sub) = fo;
                if (ok) {
                    dgo[(0 + 16 * sub) / 4] = fi;
                    dgo[(128 + 16 * sub) / 4] = ff;
                    dgo[(256 + 16 * sub) / 4] = fg;
                    dgo[(384 + 16 * sub) / 4] = fo;
                }
                ccur[rt][sub] = vcp;  // c_{t-1} is the next iteration's c_t
            }
        }
        __syncthreads();
        // ---- [dxe | dh_{t-1}]^T = Wcat^T dG^T, with next step's inputs streaming in ---------------------------------
        f32x4 acc[kSeqRT][4];
#pragma unroll
        for (int rt = 0; rt < kSeqRT; ++rt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) acc[rt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool more = t > 0;
        stream_product<32, 4, kSeqRT, kDGTile, kBwdRing>(ws, L.dg[0] + c * kDGS + 4 * g, ring, acc, [&](int gq) {
            if (gq < 4) fetch_x(gq, t);
            else if (more && gq - 4 < kSeqRT * kIn) fetch(gq - 4, t - 1);
            // gate bias gradient: row gq of the 32-row dG tile, two columns per thread (rows past R hold zeros)
            const float *drow = L.dg[gq >> 4] + (gq & 15) * kDGS + threadIdx.x;
            bcol0 += drow[0];
            bcol1 += drow[256];
        });
        // ---- outputs: dxe (with relu') to global, dh_{t-1} stays in registers ------------------------------------------
#pragma unroll
        for (int rt = 0; rt < kSeqRT; ++rt) {
            const long long row = first + 16 * rt + c;
            const bool ok = FULL || row < R;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                dhrec[rt][sub] = acc[rt][2 + sub];
                const float4 x4 = xv[rt][sub];
                f32x4 d = acc[rt][sub];
                d[0] = x4.x > 0.0f ? d[0] : 0.0f;
                d[1] = x4.y > 0.0f ? d[1] : 0.0f;
                d[2] = x4.z > 0.0f ? d[2] : 0.0f;
                d[3] = x4.w > 0.0f ? d[3] : 0.0f;
                xsum[sub] += d;
                if (ok) *reinterpret_cast<float4 *>(dxe + ((size_t)t * R + row) * kLH + uoff + 16 * sub) = make_float4(d[0], d[1], d[2], d[3]);
            }
        }
        __syncthreads();  // every wave is done with the dG tile before the next step overwrites it
    }

    // ---- per-workgroup column sums (bias gradients) ---------------------------------------------------------------------
    float *bp = bias_partial + (size_t)blockIdx.x * kBiasCols;
    bp[threadIdx.x] = bcol0;
    bp[threadIdx.x + 256] = bcol1;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s = row16_sum(xsum[sub][r]);  // over the 16 row lanes; lane c == 0 writes
            if (c == 0) bp[kLG + 16 * sub + uoff + r] = s;
        }
}

// out[col] = sum over workgroups of partial[wg][col]; cols < 512 -> gate bias gradient, the rest -> encoder bias gradient.
// One workgroup per 64 columns: 4 group-quarters x 64 columns, combined through LDS in a fixed order.
__global__ void __launch_bounds__(256) lstm_bias_final_kernel(const float *__restrict__ partial, int groups, float *__restrict__ gate_bias_grad,
                                                             float *__restrict__ enc_bias_grad) {
    __shared__ float sh[4][64];
    const int cl = threadIdx.x & 63, quarter = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    const int per = (groups + 3) / 4;
    const int lo = quarter * per, hi = lo + per < groups ? lo + per : groups;
    float s0 = 0.0f, s1 = 0.0f;
    int i = lo;
    for (; i + 2 <= hi; i += 2) {
        s0 += partial[(size_t)i * kBiasCols + col];
        s1 += partial[(size_t)(i + 1) * kBiasCols + col];
    }
    if (i < hi) s0 += partial[(size_t)i * kBiasCols + col];
    sh[quarter][cl] = s0 + s1;
    __syncthreads();
    if (quarter == 0) {
        const float s = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
        if (col < kLG) gate_bias_grad[col] = s;
        else enc_bias_grad[col - kLG] = s;
    }
}

static int check_seq_dims(const pfa_mlp_dims *d) {
    PFA_REQUIRE(d != nullptr, "lstm_seq: null dims");
    PFA_REQUIRE(d->hidden == kHidden, "lstm_seq: hidden must be %d (got %d)", kHidden, d->hidden);
    PFA_REQUIRE(d->obs_stride == 16 || d->obs_stride == 32 || d->obs_stride == 64 || d->obs_stride == 96 || d->obs_stride == 128 ||
                    d->obs_stride == 160,
                "lstm_seq: obs_stride must be one of 16/32/64/96/128/160 (got %d)", d->obs_stride);
    PFA_REQUIRE(d->num_actions >= 1 && d->num_actions <= 15, "lstm_seq: num_actions must be in 1..15 (got %d)", d->num_actions);
    return 0;
}

#define PFA_SEQ_DISPATCH_DP(dp, CALL)             \
    switch (dp) {                                 \
        case 16: { constexpr int DP = 16; CALL; } break;   \
        case 32: { constexpr int DP = 32; CALL; } break;   \
        case 64: { constexpr int DP = 64; CALL; } break;   \
        case 96: { constexpr int DP = 96; CALL; } break;   \
        case 160: { constexpr int DP = 160; CALL; } break; \
        default: { constexpr int DP = 128; CALL; } break;  \
    }

}  // namespace pfa

using namespace pfa;

// Both re-tilings of [W_ih | W_hh] an optimizer step needs (forward A fragments, csrc/lstm_fused.hip; backward A fragments) in ONE launch.
extern "C" int pfa_lstm_pack_both(const float *params, const pfa_mlp_dims *dims, void *wpack, void *wpack_bwd, pfa_stream_t stream) {
    if (int rc = check_seq_dims(dims)) return rc;
    PFA_REQUIRE(params && wpack && wpack_bwd && (uintptr_t)wpack % 16 == 0 && (uintptr_t)wpack_bwd % 16 == 0, "lstm_pack_both: bad buffer");
    hipLaunchKernelGGL(lstm_pack_bwd_kernel, dim3(2 * (kGatePackFloats / 4 / 256)), dim3(256), 0, (hipStream_t)stream, params, dims->obs_stride,
                       dims->num_actions, (float4 *)wpack_bwd, (float4 *)wpack);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_lstm_pack_bwd(const float *params, const pfa_mlp_dims *dims, void *wpack_bwd, pfa_stream_t stream) {
    if (int rc = check_seq_dims(dims)) return rc;
    PFA_REQUIRE(params && wpack_bwd && (uintptr_t)wpack_bwd % 16 == 0, "lstm_pack_bwd: bad buffer");
    hipLaunchKernelGGL(lstm_pack_bwd_kernel, dim3(kGatePackFloats / 4 / 256), dim3(256), 0, (hipStream_t)stream, params,
                       dims->obs_stride, dims->num_actions, (float4 *)wpack_bwd);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_lstm_seq_forward(const float *obs_tm, int64_t rows, int32_t steps, const float *params, const pfa_mlp_dims *dims,
                                    const void *wpack, float *xe, float *gates_act, float *hs, float *cs, int32_t init_slot, pfa_stream_t stream) {
    if (int rc = check_seq_dims(dims)) return rc;
    PFA_REQUIRE(rows >= 1 && steps >= 1, "lstm_seq_forward: empty minibatch");
    PFA_REQUIRE(obs_tm && params && wpack && xe && gates_act && hs && cs, "lstm_seq_forward: null buffer");
    PFA_REQUIRE(init_slot <= steps, "lstm_seq_forward: init_slot %d out of range (-1 = zero state, 0 .. steps)", init_slot);
    const unsigned grid = (unsigned)((rows + kSeqRows - 1) / kSeqRows);
    ScopedKernelTimer timer("lstm_seq_fwd", (hipStream_t)stream);
    PFA_SEQ_DISPATCH_DP(dims->obs_stride,
                        hipLaunchKernelGGL(lstm_seq_fwd_kernel<DP>, dim3(grid), dim3(kLstmThreads), 0, (hipStream_t)stream, obs_tm,
                                           (long long)rows, (int)steps, params, dims->num_actions, (const float4 *)wpack, xe,
                                           gates_act, hs, cs, (int)(init_slot < 0 ? -1 : init_slot)));
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t pfa_lstm_seq_backward_workspace_bytes(int64_t rows) {
    const int64_t groups = (rows + kSeqRows - 1) / kSeqRows;
    return (size_t)(groups > 0 ? groups : 1) * kBiasCols * sizeof(float);
}

extern "C" int pfa_lstm_seq_backward(const float *gates_act, const float *cs, const float *xe, const float *dh_heads, int64_t rows,
                                     int32_t steps, const void *wpack_bwd, float *dgates, float *dxe, float *gate_bias_grad,
                                     float *enc_bias_grad, void *workspace, pfa_stream_t stream) {
    PFA_REQUIRE(rows >= 1 && steps >= 1, "lstm_seq_backward: empty minibatch");
    PFA_REQUIRE(gates_act && cs && xe && dh_heads && wpack_bwd && dgates && dxe && workspace && (!gate_bias_grad == !enc_bias_grad),
                "lstm_seq_backward: null buffer");
    const unsigned grid = (unsigned)((rows + kSeqRows - 1) / kSeqRows);
    hipStream_t st = (hipStream_t)stream;
    ScopedKernelTimer timer("lstm_seq_bwd", st);
    if (rows % kSeqRows == 0)
        hipLaunchKernelGGL(lstm_seq_bwd_kernel<true>, dim3(grid), dim3(kLstmThreads), 0, st, gates_act, cs, xe, dh_heads, (long long)rows,
                           (int)steps, (const float4 *)wpack_bwd, dgates, dxe, (float *)workspace);
    else
        hipLaunchKernelGGL(lstm_seq_bwd_kernel<false>, dim3(grid), dim3(kLstmThreads), 0, st, gates_act, cs, xe, dh_heads, (long long)rows,
                           (int)steps, (const float4 *)wpack_bwd, dgates, dxe, (float *)workspace);
    PFA_LAUNCH_CHECK();
    if (!gate_bias_grad) return 0;   // the caller sums the per-workgroup column sums itself (pfa_reduce_multi, kind 1: [groups][640])
    hipLaunchKernelGGL(lstm_bias_final_kernel, dim3(kBiasCols / 64), dim3(256), 0, st, (const float *)workspace, (int)grid,
                       gate_bias_grad, enc_bias_grad);
    PFA_LAUNCH_CHECK();
    return 0;
}
