// gae.hip — Generalized Advantage Estimation as a reverse affine scan (replaces c_gae.compute_gae,
// /root/reference c_gae.pyx:11-32, called at clean_pufferl.py:168-169).
//
//   adv[t] = delta_t + coef_t * adv[t+1],   adv[n-1] = 0
//   delta_t = r[t+1] + gamma*v[t+1]*(1-d[t+1]) - v[t],   coef_t = gamma*lambda*(1-d[t+1])
//
// over the whole flat env-major batch (the recurrence deliberately runs across env boundaries, like the
// reference — SURVEY.md App. A.4).  Each element is the affine map f_t(x) = coef_t*x + delta_t; maps
// compose associatively, so:
//   pass 1  per 2048-element block: compose the block's maps (thread-serial over 8 items, wavefront
//           shuffle scan over 64 lanes, LDS over 4 waves) -> one (C, D) pair per block, in fp64;
//   pass 2  each block composes the aggregates of all LATER blocks (<= a few hundred pairs) to get its
//           carry-in, rebuilds per-thread carry-ins with the same wavefront scan, then every thread runs the
//           reference's exact fp32 recurrence (same operation order, no FMA contraction) over its own 8 items.
//           Only the carry-in is re-associated (and it is the more accurate fp64 value), so the result stays
//           within a few fp32 ulps of the sequential code: parity is allclose(atol=1e-5, rtol=1e-5).
// Roofline: HBM.  Algorithmic bytes: read r,v,d (12 B) + write adv, returns (8 B) = 20 B per element.
#include <cmath>
#include <cstdlib>

#include "common.hpp"

// Floating-point contraction is switched off where the reference's arithmetic is restated (hipcc's __fmul_rn / __fadd_rn wrappers
// contract like plain operators — checked in the ISA — so the code below uses plain operators under this pragma): every product and
// every sum of c_gae.pyx:27-30 is rounded on its own, as the reference's x86-64 build rounds them.  (Rounds 1-4 kept ONE fma, the
// recurrence's last operation, because the conv policy's update test against the double-precision oracle passed only with it; the
// cause was the few-ulp distance of ANY re-associated scan from the reference's own rounded sequence — gae_exact_kernel below
// removes that distance instead.)
#define PFA_GAE_FP _Pragma("clang fp contract(off)")

namespace pfa {

constexpr int kGaeThreads = 256;
constexpr int kGaeItems = 8;
constexpr int kGaeBlock = kGaeThreads * kGaeItems;

struct Affine {
    double c, d;  // x -> c*x + d
};

// apply `inner` first, then `outer`
__device__ __forceinline__ Affine compose(const Affine &outer, const Affine &inner) {
    return {outer.c * inner.c, outer.d + outer.c * inner.d};
}

__device__ __forceinline__ Affine shfl_down_affine(const Affine &a, int off) {
    return {__shfl_down(a.c, off, 64), __shfl_down(a.d, off, 64)};
}

// Loads the 9 values a thread needs (its 8 items + the successor) and returns the thread's composite map.
struct GaeItems {
    float v[kGaeItems + 1], r[kGaeItems + 1], d[kGaeItems + 1];
};

__device__ __forceinline__ void gae_load(const float *dones, const float *values, const float *rewards, long long s,
                                         long long n, GaeItems &it) {  // n = readable elements (incl. a halo row)
#pragma unroll
    for (int i = 0; i <= kGaeItems; ++i) {
        const long long t = s + i;
        const bool ok = t < n;
        it.v[i] = ok ? values[t] : 0.0f;
        it.r[i] = ok ? rewards[t] : 0.0f;
        it.d[i] = ok ? dones[t] : 0.0f;
    }
}

__device__ __forceinline__ Affine gae_thread_map(const GaeItems &it, long long s, long long n, float gamma, float lam,
                                                 bool pin_last) {
    PFA_GAE_FP   // (hipcc's __fmul_rn / __fadd_rn wrappers contract into fmas — checked in the ISA — so: plain operators, contraction off)
    Affine f = {1.0, 0.0};
#pragma unroll
    for (int i = kGaeItems - 1; i >= 0; --i) {
        const long long t = s + i;
        if (t >= n) continue;  // identity
        if (t == n - 1 && pin_last) {  // adv[n-1] = 0 whatever follows
            f = {0.0, 0.0};
            continue;
        }
        const float nnt = 1.0f - it.d[i + 1];
        const float delta = (it.r[i + 1] + (gamma * it.v[i + 1]) * nnt) - it.v[i];
        const float coef = (gamma * lam) * nnt;
        f = compose({(double)coef, (double)delta}, f);
    }
    return f;
}

// Inclusive suffix scan over the 256 threads of the block: on return `mine` = F_tid o F_tid+1 o ... o F_255,
// and `after` = F_tid+1 o ... o F_255 (identity for the last thread).  sh must hold 2*4 Affines.
template <int NT = kGaeThreads>
__device__ __forceinline__ void block_suffix_scan(Affine &mine, Affine &after, Affine *sh) {
    const int lane = lane_id(), wv = wave_id();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const Affine o = shfl_down_affine(mine, off);
        if (lane + off < 64) mine = compose(mine, o);
    }
    if (lane == 0) sh[wv] = mine;
    __syncthreads();
    Affine tail = {1.0, 0.0};  // composition of all later waves
    for (int q = NT / 64 - 1; q > wv; --q) tail = compose(sh[q], tail);
    mine = compose(mine, tail);
    after = shfl_down_affine(mine, 1);
    if (lane == 63) after = tail;
    __syncthreads();
}

template <int NT = kGaeThreads>
__global__ void __launch_bounds__(NT) gae_aggregate_kernel(const float *dones, const float *values,
                                                          const float *rewards, long long n, float gamma,
                                                          float lam, Affine *agg, int halo) {
    __shared__ Affine sh[NT / 64];
    const long long s = (long long)blockIdx.x * (NT * kGaeItems) + (long long)threadIdx.x * kGaeItems;
    GaeItems it;
    gae_load(dones, values, rewards, s, n + halo, it);
    Affine mine = gae_thread_map(it, s, n, gamma, lam, halo == 0), after;
    block_suffix_scan<NT>(mine, after, sh);
    if (threadIdx.x == 0) agg[blockIdx.x] = mine;
}

// The data-parallel shard form's pass 2 (pfa_gae_shard_pass2): the carry-in is the f64 composition of the later blocks, every thread
// then runs the reference's fp32 recurrence over its own 8 items from the (re-associated, f64-accurate) value behind them — within
// a few fp32 ulps of the sequential code (the start value is not the reference's own rounded one).
__global__ void __launch_bounds__(kGaeThreads) gae_apply_kernel(const float *dones, const float *values,
                                                               const float *rewards, float *adv, float *ret, long long n,
                                                               float gamma, float lam, const Affine *agg, int nblocks,
                                                               const double *carry_in, int halo) {
    __shared__ Affine sh[kGaeThreads / 64];
    __shared__ double s_carry;
    // carry-in of this block = (agg[b+1] o agg[b+2] o ... o agg[nblocks-1])(0), composed in order by wave 0
    if (wave_id() == 0) {
        const int first = blockIdx.x + 1, cnt = nblocks - first;
        const int per = (cnt + 63) / 64;
        const int lane = lane_id();
        Affine f = {1.0, 0.0};
        for (int i = per - 1; i >= 0; --i) {
            const int b = first + lane * per + i;
            if (b < nblocks) f = compose(agg[b], f);
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const Affine o = shfl_down_affine(f, off);
            if (lane + off < 64) f = compose(f, o);
        }
        // applied to x = the advantage of the element that follows this array (0 unless a later shard exists)
        if (lane == 0) s_carry = f.c * (carry_in ? *carry_in : 0.0) + f.d;
    }
    const long long s = (long long)blockIdx.x * kGaeBlock + (long long)threadIdx.x * kGaeItems;
    GaeItems it;
    gae_load(dones, values, rewards, s, n + halo, it);
    Affine mine = gae_thread_map(it, s, n, gamma, lam, halo == 0), after;
    block_suffix_scan(mine, after, sh);  // contains the __syncthreads that publishes s_carry
    float last = (float)(after.c * s_carry + after.d);  // adv at the first element after this thread's items
    {
    PFA_GAE_FP   // the reference's x86-64 build multiplies and adds separately
#pragma unroll
    for (int i = kGaeItems - 1; i >= 0; --i) {
        const long long t = s + i;
        if (t >= n) continue;
        if (t == n - 1 && halo == 0) {
            last = 0.0f;
        } else {  // the reference's statement order and rounding (c_gae.pyx:27-30)
            const float nnt = 1.0f - it.d[i + 1];
            const float delta = (it.r[i + 1] + (gamma * it.v[i + 1]) * nnt) - it.v[i];
            last = delta + ((gamma * lam) * nnt) * last;
        }
        adv[t] = last;
        if (ret) ret[t] = last + it.v[i];
    }
    }
}

// ---- single-array form: the reference's OWN fp32 sequence, in parallel (round 5) ---------------------------------------------
// c_gae.pyx:24-30 is one sequential fp32 chain from the end of the batch; a parallel scan re-associates it, and however accurate
// the re-associated start value of a thread's items is (f64 above), it is not the value the reference's chain has ROUNDED itself
// to at that point — a few ulps apart, which Adam's 1 / (|g| + eps) amplifies into the conv update's thin margin against the
// double-precision oracle.  But the recurrence is a contraction: x -> delta + coef x with
// coef <= gamma lambda < 1, and a `done` row (coef = 0) or the pinned last element restarts it exactly.  Two fp32 sequences that
// run the SAME rounded operations from start values a few ulps apart are at distance <= (gamma lambda)^k ulps after k steps and, once
// equal, stay equal.  So every thread starts W elements BEHIND its own items from the f64-accurate value there, runs the
// reference's exact fp32 statement over those W elements for nothing but its state, and arrives at its own items on the
// reference's sequence: W = ln(1e-7) / ln(gamma lambda) steps (264 at the defaults; capped at 1024, the window).  Advantages come
// out BIT-IDENTICAL to c_gae on typical data (tests/test_gpu_gae.py asserts array_equal on its fixed seeds at B = 524 288) for
// gamma lambda <= ~0.985; above that the start error has not fully died after 1024 steps and single entries may sit 1 ulp off —
// what every entry was before.  (The contraction argument holds in real arithmetic: in fp32 two sequences one ulp apart can stay
// one ulp apart for a while, so the guarantee is "the reference's bits on typical data, else within 1 ulp", and the contract the
// library states is the 1e-5 of the specification.)
//
// Layout: the array is cut into chunks of 1024 elements (pass 1: one f64 affine map per chunk).  Workgroup b owns chunk b and
// reads chunk b + 1 as its warm-up window: 256 threads x 8 elements = both chunks; all of them take part in the f64 suffix scan
// (which yields the f64-accurate advantage at every 8-element boundary of the window, carry-in = the maps of chunks b + 2 ...) and
// leave the (coef, delta) pairs of their elements in LDS; threads 0..127 then walk W + 8 elements each.  LDS index of element i is
// i + i / 8 (a lane's elements are 9 float2 apart: 2 lanes per bank instead of 16).
//
// SUMS: the same pass also leaves, per chunk, what the update and its log line need from the advantages (one pass over the
// 20 B/row stream instead of three: this kernel, adv_stats_partial_kernel, ev_partial_kernel of ppo_update.hip):
//   part[chunk][0 .. 2 nmb)   sum adv, sum adv^2 of every minibatch's rows in this chunk (clean_pufferl.py:211-213; minibatch m =
//                             segments {m + k nmb} of `bptt` rows, :455-457)
//   part[chunk][2 nmb + 0..3] sum y_true, sum y_true^2, sum adv, sum adv^2 with y_true = adv + values in STORAGE (step-major)
//                             order — the reference's mis-aligned explained-variance inputs (clean_pufferl.py:266-270, App. A.8)
// all in f64, reduced in a fixed tree (deterministic).  A thread's 8 rows lie in ONE segment (bptt % 8 == 0), thread t's minibatch
// is ((t / G) % nmb) with G = bptt / 8 threads per segment in EVERY chunk (the chunk's 128 threads are a multiple of the cycle
// P = G nmb, a power of two <= 128: gae_sums_ok).
constexpr int kGaeChunkThreads = 128;
constexpr int kGaeChunk = kGaeChunkThreads * kGaeItems;   // 1024
struct GaeSums {
    double *part;          // [chunks][2 nmb + 4]
    int nmb, bptt;         // minibatches, rows per segment
    int num_envs, horizon; // storage order j = t * num_envs + e  <->  env-major e * horizon + t
};
// SELF (gamma lambda small enough for the window to do it alone, gae_warm_self below): no f64 carry at all — neither the chunk maps
// of pass 1 (no gae_aggregate launch) nor the block scan.  Every walker starts from x = 0 at `warm` elements behind its items: the
// start error is then the advantage itself instead of a few of its ulps, and 24 more binary orders of contraction
// ((gamma lambda)^warm <= 1e-7 * 2^-24) bring it as far below one ulp as the f64-seeded form is after its shorter warm-up — 536
// elements instead of 264 at the defaults, inside the 1024-element window for gamma lambda <= 0.968 (CT = 128), inside the 2048-element one up to 0.984 (CT = 256).  Beyond the end of the array
// the window holds (0, 0) pairs, where x = 0 IS the reference's value (adv[n-1] = 0, c_gae.pyx:24).
// CT = walker threads = chunk / 8: 128 (chunks and windows of 1024 elements: every form) or, self-starting form only, 256 (2048: the
// warm-up of gamma lambda up to 0.984 fits the window, so those discounts also run as ONE launch, and sharded over ranks as the halo form)
template <bool SUMS, bool SELF, int CT = kGaeChunkThreads>
__global__ void __launch_bounds__(2 * CT) gae_exact_kernel(const float *dones, const float *values, const float *rewards, float *adv,
                                                               float *ret, long long n, long long n_read, float gamma, float lam,
                                                               const Affine *agg, int nchunks, int warm, GaeSums sums) {
    constexpr int kChunk = CT * kGaeItems, kWin = 2 * kChunk, kNT = 2 * CT;
    __shared__ Affine sh[SELF ? 1 : kNT / 64];
    __shared__ double s_carry;
    __shared__ float2 cd[kWin + kWin / 8];      // (coef, delta) of the window's elements, index i + i / 8
    __shared__ float start[SELF ? 1 : kNT];       // f64-accurate advantage at element 8 (t + 1) of the window
    __shared__ double sh_s[SUMS ? 4 : 1][SUMS ? CT : 1];
    const int tid = threadIdx.x;
    // carry-in of the WINDOW (chunks b, b + 1) = (agg[b+2] o ... o agg[nchunks-1])(0), composed in order by wave 0
    if (!SELF && wave_id() == 0) {
        const int first = blockIdx.x + 2, cnt = nchunks - first;
        const int per = cnt > 0 ? (cnt + 63) / 64 : 0;
        const int lane = lane_id();
        Affine f = {1.0, 0.0};
        for (int i = per - 1; i >= 0; --i) {
            const int b = first + lane * per + i;
            if (b < nchunks) f = compose(agg[b], f);
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const Affine o = shfl_down_affine(f, off);
            if (lane + off < 64) f = compose(f, o);
        }
        if (lane == 0) s_carry = f.d;   // applied to 0: nothing follows the array
    }
    const long long s = (long long)blockIdx.x * kChunk + (long long)tid * kGaeItems;
    GaeItems it;
    // SELF: only the window elements a walker can reach are fetched (the last walker starts at element 1024 + 8 + warm)
    const bool feeds = !SELF || tid * kGaeItems < kChunk + kGaeItems + warm;
    if (feeds) gae_load(dones, values, rewards, s, n_read, it);
    if constexpr (!SELF) {
        Affine mine = gae_thread_map(it, s, n, gamma, lam, true), after;
        block_suffix_scan<kNT>(mine, after, sh);  // contains the __syncthreads that publishes s_carry
        start[tid] = (float)(after.c * s_carry + after.d);
    }
    if (feeds) {
    PFA_GAE_FP   // the reference's x86-64 build multiplies and adds separately (c_gae.pyx:27-30)
#pragma unroll
    for (int i = 0; i < kGaeItems; ++i) {
        const long long t = s + i;
        float2 e = make_float2(0.0f, 0.0f);           // t >= n_read - 1: adv = 0 (the pinned last element; nothing beyond the array)
        if (t < n_read - 1) {
            const float nnt = 1.0f - it.d[i + 1];
            e.y = (it.r[i + 1] + (gamma * it.v[i + 1]) * nnt) - it.v[i];
            e.x = (gamma * lam) * nnt;
        }
        const int li = tid * kGaeItems + i;
        cd[li + (li >> 3)] = e;
    }
    }
    __syncthreads();
    float advv[kGaeItems];
#pragma unroll
    for (int i = 0; i < kGaeItems; ++i) advv[i] = 0.0f;
    if (tid < CT) {
        PFA_GAE_FP
        const int first = tid * kGaeItems + kGaeItems;          // window element behind this thread's items
        float x = SELF ? 0.0f : start[tid + (warm >> 3)];       // advantage at window element first + warm (8-aligned); SELF: see above
        // warm-up: the reference's statement, for the state only.  Groups of 8 elements (warm and first are multiples of 8, so a
        // group is 8 consecutive float2 in LDS): the group's reads are issued together, then its 8 dependent steps — left as one
        // read per step the loop is a chain of LDS latencies (28.8 us per launch measured; the arithmetic is 2 instructions a step)
        for (int li = first + warm - kGaeItems; li >= first; li -= kGaeItems) {
            const float2 *grp = cd + li + (li >> 3);
            float2 e[kGaeItems];
#pragma unroll
            for (int k = 0; k < kGaeItems; ++k) e[k] = grp[k];
#pragma unroll
            for (int k = kGaeItems - 1; k >= 0; --k) x = e[k].y + e[k].x * x;
        }
#pragma unroll
        for (int i = kGaeItems - 1; i >= 0; --i) {
            const int li = tid * kGaeItems + i;
            const float2 e = cd[li + (li >> 3)];
            x = e.y + e.x * x;                                  // lastgaelam = delta + gamma * gae_lambda * nextnonterminal * lastgaelam
            advv[i] = x;
            const long long t = s + i;
            if (t < n) {
                adv[t] = x;
                if (ret) ret[t] = x + it.v[i];
            }
        }
    }
    if constexpr (SUMS) {
        double a1 = 0.0, a2 = 0.0, y1 = 0.0, y2 = 0.0;
        if (tid < CT) {
            // y_pred of flat index j is values[(j % N) * T + j / N]: issue the thread's eight gathers together
            float yp[kGaeItems];
            const long long e0 = s % sums.num_envs, t0 = s / sums.num_envs;
#pragma unroll
            for (int i = 0; i < kGaeItems; ++i) {
                long long e = e0 + i, t = t0;
                while (e >= sums.num_envs) {   // (at most once unless num_envs < 8)
                    e -= sums.num_envs;
                    ++t;
                }
                yp[i] = s + i < n ? values[e * sums.horizon + t] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < kGaeItems; ++i) {
                if (s + i >= n) continue;
                const double a = (double)advv[i], y = a + (double)yp[i];
                a1 += a;
                a2 += a * a;
                y1 += y;
                y2 += y * y;
            }
            sh_s[0][tid] = a1;
            sh_s[1][tid] = a2;
            sh_s[2][tid] = y1;
            sh_s[3][tid] = y2;
        }
        __syncthreads();
        const int G = sums.bptt / kGaeItems, P = G * sums.nmb;   // threads per segment, per cycle of minibatches
        // fixed tree down to one entry per (minibatch, position in the segment): t and t + stride share both while stride >= P
        for (int stride = CT / 2; stride >= P; stride >>= 1) {
            if (tid < stride) {
#pragma unroll
                for (int q = 0; q < 4; ++q) sh_s[q][tid] += sh_s[q][tid + stride];
            }
            __syncthreads();
        }
        double *out = sums.part + (size_t)blockIdx.x * (2 * sums.nmb + 4);
        if (tid < 2 * sums.nmb) {          // thread (m, k): sum of statistic k over the G entries of minibatch m, in order
            const int m = tid >> 1, k = tid & 1;
            double v = 0.0;
            for (int g = 0; g < G; ++g) v += sh_s[k][m * G + g];
            out[tid] = v;
        } else if (tid >= 64 && tid < 68) {   // the four whole-chunk sums, entries 0 .. P-1 in order
            const int k = tid - 64, src = k < 2 ? 2 + k : k - 2;   // out: y1, y2, a1, a2
            double v = 0.0;
            for (int g = 0; g < P; ++g) v += sh_s[src][g];
            out[2 * sums.nmb + k] = v;
        }
    }
}

// Sum of the per-block pieces in a fixed order: one 64-lane workgroup per output (2 nmb + 4 of them).
// adv_stats[2 m + k] <- minibatch sums; ev4[0..3] <- sum y_true, sum y_true^2, sum adv, sum adv^2; zero8 (optional): 8 doubles
// cleared here (the update's running loss sums, clean_pufferl.train's loss_acc — saves its own fill launch).
__global__ void __launch_bounds__(64) gae_sums_final_kernel(const double *part, int nblocks, int nmb, double *adv_stats, double *ev4, double *zero8) {
    const int q = blockIdx.x, lane = threadIdx.x, width = 2 * nmb + 4;
    double v = 0.0;
    for (int b = lane; b < nblocks; b += 64) v += part[(size_t)b * width + q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) {
        if (q < 2 * nmb) adv_stats[q] = v;
        else ev4[q - 2 * nmb] = v;
    }
    if (q == 0 && zero8 && lane < 8) zero8[lane] = 0.0;
}

// ---- data-parallel GAE with ONE exchange -------------------------------------------------------------------------------------
// A shard's map is (interior) o (last element), and only the last element's map needs the next shard's first row.  So every rank
// publishes six numbers that come from its OWN rows alone — the interior map (C, D) of elements 0 .. n-2, values[n-1], and its
// first row (done, value, reward) — and after one all-gather of those every rank can finish all the shards' maps itself.
// Publish: out[0 .. n_extra) = extra (the caller's other f64 sums, e.g. the episode statistics, so that they ride the same
// all-reduce), out[n_extra + 6 q + j] = this rank's six numbers for q == rank and 0 elsewhere (all-gather as a sum).
__global__ void gae_shard_publish_kernel(const Affine *agg, int nblocks, const float *dones, const float *values, const float *rewards,
                                         long long n, const double *extra, int n_extra, double *out, int rank, int world) {
    const int lane = lane_id();
    const int per = (nblocks + 63) / 64;
    Affine f = {1.0, 0.0};
    for (int i = per - 1; i >= 0; --i) {
        const int b = lane * per + i;
        if (b < nblocks) f = compose(agg[b], f);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const Affine o = shfl_down_affine(f, off);
        if (lane + off < 64) f = compose(f, o);
    }
    const double c = __shfl(f.c, 0, 64), d = __shfl(f.d, 0, 64);
    for (int i = lane; i < n_extra; i += 64) out[i] = extra[i];
    for (int i = lane; i < 6 * world; i += 64) {
        double v = 0.0;
        if (i / 6 == rank) {
            const int j = i % 6;
            v = j == 0 ? c : j == 1 ? d : j == 2 ? (double)values[n - 1] : j == 3 ? (double)dones[0] : j == 4 ? (double)values[0] : (double)rewards[0];
        }
        out[n_extra + i] = v;
    }
}

// Fold (one thread): pub = the gathered [world][6].  Completes shard q's map with its last element (the reference's statement
// order and rounding, c_gae.pyx:27-30, on the next shard's first row; the very last element of the batch is pinned to adv = 0),
// folds the later shards into this rank's carry-in, patches the last block aggregate of THIS shard (pass 2's apply kernel composes
// the aggregates of the later blocks) and drops the next shard's first row behind the shard's arrays as halo.
__global__ void gae_shard_fold_kernel(const double *pub, int rank, int world, long long n, float gamma, float lam, Affine *agg, int nblocks,
                                      float *dones, float *values, float *rewards, double *carry_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    auto last_map = [&](int q) -> Affine {
        PFA_GAE_FP
        if (q == world - 1) return Affine{0.0, 0.0};
        const float d1 = (float)pub[6 * (q + 1) + 3], v1 = (float)pub[6 * (q + 1) + 4], r1 = (float)pub[6 * (q + 1) + 5];
        const float v0 = (float)pub[6 * q + 2];
        const float nnt = 1.0f - d1;
        const float delta = (r1 + (gamma * v1) * nnt) - v0;
        const float coef = (gamma * lam) * nnt;
        return Affine{(double)coef, (double)delta};
    };
    double x = 0.0;
    for (int q = world - 1; q > rank; --q) {
        const Affine m = compose(Affine{pub[6 * q], pub[6 * q + 1]}, last_map(q));
        x = m.c * x + m.d;
    }
    carry_out[0] = x;
    agg[nblocks - 1] = compose(agg[nblocks - 1], last_map(rank));
    if (rank < world - 1) {
        dones[n] = (float)pub[6 * (rank + 1) + 3];
        values[n] = (float)pub[6 * (rank + 1) + 4];
        rewards[n] = (float)pub[6 * (rank + 1) + 5];
    }
}

// ---- data-parallel GAE, halo form: the flat scan's own bits on every rank (round 6) ------------------------------------------
// The self-starting window (gae_exact_kernel<.., SELF>) needs nothing from the rows behind a walker's items but the rows
// themselves: `warm` of them.  For a shard of the rank-major flat batch those rows are its own, except behind its last items, where
// they are the first rows of the NEXT shard(s).  So every rank publishes its first min(n, H) rows (done, value, reward),
// H = warm + 8, one all-reduce gathers them (next to the episode statistics), every rank drops the H rows that follow its shard
// behind its arrays and runs the SAME kernel the single-rank path runs with n_read = n + halo: every walker is then on the
// reference's own rounded sequence exactly as in the flat scan, shard boundary or not (tests/test_gpu_dp.py: array_equal at 8 ranks).
// The rows travel as their BIT PATTERNS in f64 (an integer < 2^32 plus zeros from the other ranks sums exactly; a float sum would
// turn -0.0 into +0.0).  Layout of the exchange buffer: out[0 .. n_extra) = extra, out[n_extra + (3 q + k) hp + i] = bits of
// {dones, values, rewards}[k][i] of rank q, hp = min(n, H).
__global__ void __launch_bounds__(256) gae_halo_publish_kernel(const float *dones, const float *values, const float *rewards, int hp,
                                                               const double *extra, int n_extra, double *out, int rank, int world) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n_extra) {
        out[i] = extra[i];
        return;
    }
    const long long j = i - n_extra;
    if (j >= (long long)world * 3 * hp) return;
    const int q = (int)(j / (3 * hp)), k = (int)(j % (3 * hp)) / hp, row = (int)(j % hp);
    double v = 0.0;
    if (q == rank) v = (double)__float_as_uint((k == 0 ? dones : k == 1 ? values : rewards)[row]);
    out[i] = v;
}

// Halo element i of rank `rank` = row (i % n) of rank rank + 1 + i / n (all shards hold n rows); halo_len <= (world - 1 - rank) n.
__global__ void __launch_bounds__(256) gae_halo_unpack_kernel(const double *gathered, int hp, int rank, long long n, int halo_len, float *dones,
                                                              float *values, float *rewards) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= halo_len) return;
    const int q = rank + 1 + (int)(i / n), row = (int)(i % n);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const unsigned bits = (unsigned)gathered[((size_t)3 * q + k) * hp + row];
        (k == 0 ? dones : k == 1 ? values : rewards)[n + i] = __uint_as_float(bits);
    }
}

}  // namespace pfa

using namespace pfa;

extern "C" size_t pfa_gae_workspace_bytes(int64_t n) {   // one f64 affine map per 1024-element chunk (covers the shard forms' 2048-element blocks)
    const int64_t nc = (n + kGaeChunk - 1) / kGaeChunk;
    return (size_t)(nc > 0 ? nc : 1) * sizeof(Affine);
}

// Warm-up elements of gae_exact_kernel: the start error (a few ulps) times (gamma lambda)^W must be far below one ulp.
static int gae_warm(float gamma, float lam) {
    const double gl = (double)(gamma * lam);
    if (!(gl > 0.0)) return 8;                       // coef = 0 everywhere: every element restarts the chain
    if (gl >= 0.999) return kGaeChunk;
    const double w = std::ceil(std::log(1e-7) / std::log(gl) / 8.0) * 8.0;
    return (int)(w < 8.0 ? 8.0 : (w > (double)kGaeChunk ? (double)kGaeChunk : w));
}

// Warm-up elements of the self-starting form (gae_exact_kernel<.., SELF>): the start error is the advantage itself, so the
// contraction has to cover the 24 bits of an fp32 mantissa on top of gae_warm's margin.  0 = even the 2048-element window is too
// short for this gamma lambda (> 0.984): the f64-seeded two-launch form runs.  PFA_GAE_SELF=0 forces the latter (A/B, fallback).
static int gae_warm_self(float gamma, float lam) {
    const char *e = std::getenv("PFA_GAE_SELF");   // (read per call: the tests flip it inside one process)
    if (e && e[0] == '0') return 0;
    const double gl = (double)(gamma * lam);
    if (!(gl > 0.0)) return 8;
    if (gl >= 0.999) return 0;
    const double w = std::ceil(std::log(1e-7 * 5.9604644775390625e-08) / std::log(gl) / 8.0) * 8.0;
    return w <= 2.0 * kGaeChunk ? (int)(w < 8.0 ? 8.0 : w) : 0;   // (<= 1024: chunks of 1024; <= 2048, gamma lambda <= 0.984: chunks of 2048)
}

// The self-starting form for `ws` warm-up elements over n rows reading n_read (>= n: the data-parallel halo); returns its chunk count.
template <bool SUMS>
static int64_t launch_gae_self(const float *dones, const float *values, const float *rewards, float *advantages, float *returns, int64_t n,
                               int64_t n_read, float gamma, float gae_lambda, int ws, GaeSums sums, hipStream_t st) {
    if (ws <= kGaeChunk) {
        const int64_t nc = (n + kGaeChunk - 1) / kGaeChunk;
        hipLaunchKernelGGL((gae_exact_kernel<SUMS, true, kGaeChunkThreads>), dim3((unsigned)nc), dim3(2 * kGaeChunkThreads), 0, st, dones, values, rewards,
                           advantages, returns, (long long)n, (long long)n_read, gamma, gae_lambda, (const Affine *)nullptr, (int)nc, ws, sums);
        return nc;
    }
    const int64_t nc = (n + 2 * kGaeChunk - 1) / (2 * kGaeChunk);
    hipLaunchKernelGGL((gae_exact_kernel<SUMS, true, 2 * kGaeChunkThreads>), dim3((unsigned)nc), dim3(4 * kGaeChunkThreads), 0, st, dones, values, rewards,
                       advantages, returns, (long long)n, (long long)n_read, gamma, gae_lambda, (const Affine *)nullptr, (int)nc, ws, sums);
    return nc;
}

extern "C" int pfa_gae_f32(const float *dones, const float *values, const float *rewards, float *advantages, float *returns,
                           int64_t n, float gamma, float gae_lambda, void *workspace, pfa_stream_t stream) {
    PFA_REQUIRE(n >= 0, "gae: negative length");
    if (n == 0) return 0;
    PFA_REQUIRE(dones && values && rewards && advantages && workspace, "gae: null buffer");
    const int64_t nc = (n + kGaeChunk - 1) / kGaeChunk;
    PFA_REQUIRE(nc <= 0x7fffffff, "gae: batch too large");
    Affine *agg = (Affine *)workspace;
    ScopedKernelTimer timer("gae", (hipStream_t)stream);  // both passes
    if (const int ws = gae_warm_self(gamma, gae_lambda)) {   // one launch: the window warms itself up (no chunk maps)
        launch_gae_self<false>(dones, values, rewards, advantages, returns, n, n, gamma, gae_lambda, ws, GaeSums{}, (hipStream_t)stream);
        PFA_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(gae_aggregate_kernel<kGaeChunkThreads>, dim3((unsigned)nc), dim3(kGaeChunkThreads), 0, (hipStream_t)stream, dones, values,
                       rewards, (long long)n, gamma, gae_lambda, agg, 0);
    PFA_LAUNCH_CHECK();
    hipLaunchKernelGGL((gae_exact_kernel<false, false>), dim3((unsigned)nc), dim3(kGaeThreads), 0, (hipStream_t)stream, dones, values, rewards,
                       advantages, returns, (long long)n, (long long)n, gamma, gae_lambda, agg, (int)nc, gae_warm(gamma, gae_lambda), GaeSums{});
    PFA_LAUNCH_CHECK();
    return 0;
}

// compute_gae + the advantage statistics of the update in one pass (see GaeSums): what pfa_gae_f32, pfa_ppo_adv_stats and the
// sums of pfa_train_log_sums leave, in three launches instead of six.  Supported when a thread's 8 rows never straddle a segment
// and the minibatch cycle divides the chunk (pfa_gae_sums_supported); the caller falls back to the separate entry points otherwise.
static bool gae_sums_ok(int64_t n, int num_envs, int nmb, int bptt) {
    if (n < 1 || num_envs < 1 || nmb < 1 || bptt < kGaeItems || bptt % kGaeItems) return false;
    if (n % num_envs || n % ((int64_t)nmb * bptt)) return false;
    const int64_t P = (int64_t)(bptt / kGaeItems) * nmb;
    return P <= kGaeChunkThreads && (P & (P - 1)) == 0 && 2 * nmb <= 64;
}
extern "C" int pfa_gae_sums_supported(int64_t n, int32_t num_envs, int32_t num_minibatches, int32_t bptt_horizon) {
    return gae_sums_ok(n, num_envs, num_minibatches, bptt_horizon) ? 1 : 0;
}
extern "C" size_t pfa_gae_sums_workspace_bytes(int64_t n, int32_t num_minibatches) {
    const int64_t nc = (n + kGaeChunk - 1) / kGaeChunk;
    return align_up((size_t)(nc > 0 ? nc : 1) * sizeof(Affine), 256) + (size_t)(nc > 0 ? nc : 1) * (2 * (size_t)num_minibatches + 4) * sizeof(double);
}
extern "C" int pfa_gae_sums_f32(const float *dones, const float *values, const float *rewards, float *advantages, float *returns,
                                int64_t n, float gamma, float gae_lambda, int32_t num_envs, int32_t num_minibatches, int32_t bptt_horizon,
                                double *adv_stats, double *ev4, double *zero8, void *workspace, pfa_stream_t stream) {
    PFA_REQUIRE(gae_sums_ok(n, num_envs, num_minibatches, bptt_horizon), "gae_sums: unsupported partition (pfa_gae_sums_supported)");
    PFA_REQUIRE(dones && values && rewards && advantages && adv_stats && ev4 && workspace, "gae_sums: null buffer");
    const int64_t nc = (n + kGaeChunk - 1) / kGaeChunk;
    PFA_REQUIRE(nc <= 0x7fffffff, "gae: batch too large");
    Affine *agg = (Affine *)workspace;
    GaeSums sums{(double *)((char *)workspace + align_up((size_t)nc * sizeof(Affine), 256)), num_minibatches, bptt_horizon, num_envs,
                 (int)(n / num_envs)};
    ScopedKernelTimer timer("gae", (hipStream_t)stream);  // all launches (three; two in the self-starting form)
    int64_t nparts = nc;
    if (const int ws = gae_warm_self(gamma, gae_lambda)) {
        nparts = launch_gae_self<true>(dones, values, rewards, advantages, returns, n, n, gamma, gae_lambda, ws, sums, (hipStream_t)stream);
    } else {
        hipLaunchKernelGGL(gae_aggregate_kernel<kGaeChunkThreads>, dim3((unsigned)nc), dim3(kGaeChunkThreads), 0, (hipStream_t)stream, dones, values,
                           rewards, (long long)n, gamma, gae_lambda, agg, 0);
        PFA_LAUNCH_CHECK();
        hipLaunchKernelGGL((gae_exact_kernel<true, false>), dim3((unsigned)nc), dim3(kGaeThreads), 0, (hipStream_t)stream, dones, values, rewards,
                           advantages, returns, (long long)n, (long long)n, gamma, gae_lambda, agg, (int)nc, gae_warm(gamma, gae_lambda), sums);
    }
    PFA_LAUNCH_CHECK();
    hipLaunchKernelGGL(gae_sums_final_kernel, dim3((unsigned)(2 * num_minibatches + 4)), dim3(64), 0, (hipStream_t)stream, sums.part, (int)nparts,
                       (int)num_minibatches, adv_stats, ev4, zero8);
    PFA_LAUNCH_CHECK();
    return 0;
}

// Pass 2 of the f64-carry shard form (see pfa_gae_shard_publish below): has_next != 0 means a later shard exists — the arrays then
// hold n+1 readable elements, element n being the next shard's first row (its done / value / reward enter this shard's last
// delta), and the last element is an interior row instead of the pinned adv = 0.
extern "C" int pfa_gae_shard_pass2(const float *dones, const float *values, const float *rewards, float *advantages,
                                   float *returns, int64_t n, int has_next, float gamma, float gae_lambda,
                                   const void *workspace, const double *carry_in, pfa_stream_t stream) {
    PFA_REQUIRE(n >= 1 && dones && values && rewards && advantages && workspace, "gae_shard: bad arguments");
    PFA_REQUIRE(!has_next || carry_in, "gae_shard: a shard with a successor needs carry_in");
    const int64_t nb = (n + kGaeBlock - 1) / kGaeBlock;
    hipLaunchKernelGGL(gae_apply_kernel, dim3((unsigned)nb), dim3(kGaeThreads), 0, (hipStream_t)stream, dones, values, rewards,
                       advantages, returns, (long long)n, gamma, gae_lambda, (const Affine *)workspace, (int)nb,
                       has_next ? carry_in : (const double *)nullptr, has_next ? 1 : 0);
    PFA_LAUNCH_CHECK();
    return 0;
}

// One-exchange form of the data-parallel scan (see gae_shard_publish_kernel): publish -> ONE all-reduce(SUM) of
// out[n_extra + 6 world] by the caller -> fold -> pfa_gae_shard_pass2 (has_next = rank < world - 1, carry_in = carry_out).
// The arrays hold n + 1 elements (the fold writes the halo row at index n); `workspace` must stay untouched between the three.
extern "C" int pfa_gae_shard_publish(const float *dones, const float *values, const float *rewards, int64_t n, float gamma, float gae_lambda,
                                     void *workspace, const double *extra, int32_t n_extra, double *out, int32_t rank, int32_t world,
                                     pfa_stream_t stream) {
    PFA_REQUIRE(n >= 1 && dones && values && rewards && workspace && out && world >= 1 && rank >= 0 && rank < world && n_extra >= 0 &&
                    (n_extra == 0 || extra),
                "gae_shard_publish: bad arguments");
    const int64_t nb = (n + kGaeBlock - 1) / kGaeBlock;
    PFA_REQUIRE(nb <= 0x7fffffff, "gae: batch too large");
    Affine *agg = (Affine *)workspace;
    // interior map: elements 0 .. n-2 (each reads its successor, element n-1 at most); the blocks are those of the n-element pass 2
    hipLaunchKernelGGL(gae_aggregate_kernel<kGaeThreads>, dim3((unsigned)nb), dim3(kGaeThreads), 0, (hipStream_t)stream, dones, values, rewards,
                       (long long)(n - 1), gamma, gae_lambda, agg, 1);
    PFA_LAUNCH_CHECK();
    hipLaunchKernelGGL(gae_shard_publish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, agg, (int)nb, dones, values, rewards, (long long)n,
                       extra, (int)n_extra, out, (int)rank, (int)world);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_gae_shard_fold(const double *gathered, int32_t rank, int32_t world, int64_t n, float gamma, float gae_lambda, void *workspace,
                                  float *dones, float *values, float *rewards, double *carry_out, pfa_stream_t stream) {
    PFA_REQUIRE(gathered && workspace && dones && values && rewards && carry_out && n >= 1 && world >= 1 && rank >= 0 && rank < world,
                "gae_shard_fold: bad arguments");
    const int64_t nb = (n + kGaeBlock - 1) / kGaeBlock;
    hipLaunchKernelGGL(gae_shard_fold_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, gathered, (int)rank, (int)world, (long long)n, gamma,
                       gae_lambda, (Affine *)workspace, (int)nb, dones, values, rewards, carry_out);
    PFA_LAUNCH_CHECK();
    return 0;
}

// Halo form of the data-parallel scan (see gae_halo_publish_kernel).  pfa_gae_halo_rows: H for this gamma lambda, 0 when the
// self-starting window cannot serve it (gamma lambda > 0.984, or PFA_GAE_SELF=0): the f64-carry form above runs then.
//   publish  out[n_extra + 3 world min(n, H)] <- extra | this rank's first rows as bit patterns (zeros in the other ranks' places)
//   (the caller all-reduces `out`, SUM)
//   unpack   gathered = out + n_extra: rows n .. n + halo_len of the three arrays <- the rows that follow this shard in the
//            rank-major flat batch, halo_len = min(H, (world - 1 - rank) n) (returned); the arrays hold n + H elements
//   pfa_gae_halo_f32   the scan over n rows reading n + halo_len; adv_stats != NULL: + the update's sums (pfa_gae_sums_f32's, over
//            this rank's rows: the caller all-reduces them)
extern "C" int32_t pfa_gae_halo_rows(float gamma, float gae_lambda) {
    const int ws = gae_warm_self(gamma, gae_lambda);
    return ws ? ws + kGaeItems : 0;
}
static int gae_halo_len(int64_t n, int32_t rank, int32_t world, int H) {
    const int64_t follow = (int64_t)(world - 1 - rank) * n;
    return (int)(follow < H ? follow : H);
}
extern "C" int pfa_gae_halo_publish(const float *dones, const float *values, const float *rewards, int64_t n, float gamma, float gae_lambda,
                                    const double *extra, int32_t n_extra, double *out, int32_t rank, int32_t world, pfa_stream_t stream) {
    const int H = pfa_gae_halo_rows(gamma, gae_lambda);
    PFA_REQUIRE(H > 0, "gae_halo: gamma * lambda = %g is outside the self-starting window (pfa_gae_halo_rows)", (double)(gamma * gae_lambda));
    PFA_REQUIRE(n >= 1 && dones && values && rewards && out && world >= 1 && rank >= 0 && rank < world && n_extra >= 0 && (n_extra == 0 || extra),
                "gae_halo_publish: bad arguments");
    const int hp = (int)(n < H ? n : H);
    const long long total = (long long)n_extra + (long long)world * 3 * hp;
    hipLaunchKernelGGL(gae_halo_publish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dones, values, rewards, hp,
                       extra, (int)n_extra, out, (int)rank, (int)world);
    PFA_LAUNCH_CHECK();
    return 0;
}
extern "C" int pfa_gae_halo_unpack(const double *gathered, int32_t rank, int32_t world, int64_t n, float gamma, float gae_lambda, float *dones,
                                   float *values, float *rewards, pfa_stream_t stream) {
    const int H = pfa_gae_halo_rows(gamma, gae_lambda);
    PFA_REQUIRE(H > 0, "gae_halo: gamma * lambda = %g is outside the self-starting window (pfa_gae_halo_rows)", (double)(gamma * gae_lambda));
    PFA_REQUIRE(gathered && dones && values && rewards && n >= 1 && world >= 1 && rank >= 0 && rank < world, "gae_halo_unpack: bad arguments");
    const int len = gae_halo_len(n, rank, world, H);
    if (len > 0) {
        hipLaunchKernelGGL(gae_halo_unpack_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gathered,
                           (int)(n < H ? n : H), (int)rank, (long long)n, len, dones, values, rewards);
        PFA_LAUNCH_CHECK();
    }
    return len;
}
extern "C" int pfa_gae_halo_f32(const float *dones, const float *values, const float *rewards, float *advantages, float *returns, int64_t n,
                                int32_t halo_len, float gamma, float gae_lambda, int32_t num_envs, int32_t num_minibatches, int32_t bptt_horizon,
                                double *adv_stats, double *ev4, double *zero8, void *workspace, pfa_stream_t stream) {
    const int ws = gae_warm_self(gamma, gae_lambda);
    PFA_REQUIRE(ws > 0, "gae_halo: gamma * lambda = %g is outside the self-starting window (pfa_gae_halo_rows)", (double)(gamma * gae_lambda));
    PFA_REQUIRE(n >= 1 && halo_len >= 0 && halo_len <= ws + kGaeItems && dones && values && rewards && advantages, "gae_halo: bad arguments");
    const int64_t nc = (n + kGaeChunk - 1) / kGaeChunk;
    PFA_REQUIRE(nc <= 0x7fffffff, "gae: batch too large");
    ScopedKernelTimer timer("gae", (hipStream_t)stream);
    if (!adv_stats) {
        launch_gae_self<false>(dones, values, rewards, advantages, returns, n, n + halo_len, gamma, gae_lambda, ws, GaeSums{}, (hipStream_t)stream);
        PFA_LAUNCH_CHECK();
        return 0;
    }
    PFA_REQUIRE(gae_sums_ok(n, num_envs, num_minibatches, bptt_horizon), "gae_halo: unsupported partition for the sums (pfa_gae_sums_supported)");
    PFA_REQUIRE(ev4 && workspace, "gae_halo: null buffer");
    GaeSums sums{(double *)((char *)workspace + align_up((size_t)nc * sizeof(Affine), 256)), num_minibatches, bptt_horizon, num_envs,
                 (int)(n / num_envs)};
    const int64_t nparts = launch_gae_self<true>(dones, values, rewards, advantages, returns, n, n + halo_len, gamma, gae_lambda, ws, sums, (hipStream_t)stream);
    PFA_LAUNCH_CHECK();
    hipLaunchKernelGGL(gae_sums_final_kernel, dim3((unsigned)(2 * num_minibatches + 4)), dim3(64), 0, (hipStream_t)stream, sums.part, (int)nparts,
                       (int)num_minibatches, adv_stats, ev4, zero8);
    PFA_LAUNCH_CHECK();
    return 0;
}
