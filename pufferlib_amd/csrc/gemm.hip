// gemm.hip — fp32 MFMA contractions over the ROW dimension of two row-major activations,
//   C[mo][no] = sum_k A[k][mo] * B[k][no]        (C = A^T B, k = minibatch rows, 10^5..10^6 of them)
// i.e. the weight-gradient products autograd forms for nn.Linear / nn.LSTM in the reference's
// loss.backward() (clean_pufferl.py:244): dW_ih = dG^T xe, dW_hh = dG^T h_prev, dW_enc = dxe^T obs,
// dW_heads = dout^T h.  The output is tiny (<= 512 x 128) and k is huge, so the work is split over k:
//   pass 1  WG (tile, split) owns a TM x TN output tile and a contiguous range of k; both operands are staged
//           k-major through a double-buffered LDS slab (BK = 16 rows) and consumed as v_mfma_f32_16x16x4_f32
//           fragments (A-op lane (c,g) = A[4kk+g][i0+c], B-op = B[4kk+g][j0+c]); row stride TM+16 floats keeps the
//           fragment reads conflict-free (bank = 16g + c within a 32-lane half);
//   pass 2  deterministic sum of the split partials in f64 -> C.
// Roofline: fp32 MFMA (2*mo*no flop per k row) for the wide products, HBM (reading A and B once) for the
// 16-row head product.
#include "common.hpp"
#include "mlp_tile.hpp"

namespace pfa {

constexpr int kGemmBK = 16;
constexpr int kGemmThreads = 256;

template <int WR, int WC, int MI, int NI>
struct GemmTnCfg {
    static constexpr int TM = WR * MI * 16, TN = WC * NI * 16;
    // LDS row strides (floats), == 16 mod 32
    static constexpr int SA = TM % 32 == 0 ? TM + 16 : TM + 32, SB = TN % 32 == 0 ? TN + 16 : TN + 32;
    static constexpr int kStageFloats = kGemmBK * (SA + SB);
    static constexpr int kA4 = kGemmBK * TM / 4, kB4 = kGemmBK * TN / 4;  // float4 per stage
    static constexpr int kLA = (kA4 + kGemmThreads - 1) / kGemmThreads, kLB = (kB4 + kGemmThreads - 1) / kGemmThreads;
    static_assert(WR * WC * 64 == kGemmThreads, "4 waves");
};

template <int WR, int WC, int MI, int NI>
__global__ void __launch_bounds__(kGemmThreads, 2)
    gemm_tn_partial_kernel(const float *__restrict__ A, long long lda, const float *__restrict__ B, long long ldb, int mo, int no,
                           long long K, long long k_per_split, float *__restrict__ partial, const float *__restrict__ B2 = nullptr,
                           long long ldb2 = 0, int nb = 0) {   // B2 != null: output columns >= nb come from B2[k][col - nb] (two-operand form)
    using Cfg = GemmTnCfg<WR, WC, MI, NI>;
    __shared__ float lds[2 * Cfg::kStageFloats];
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
    const int c = lane & 15, g = lane >> 4;
    const int wr = wv / WC, wc = wv % WC;
    const int tiles_n = no / Cfg::TN;
    const int tile = blockIdx.x, split = blockIdx.y;
    const int i0 = (tile / tiles_n) * Cfg::TM, j0 = (tile % tiles_n) * Cfg::TN;
    const long long k_lo = (long long)split * k_per_split;
    const long long k_hi = k_lo + k_per_split < K ? k_lo + k_per_split : K;
    const int stages = (int)((k_hi - k_lo + kGemmBK - 1) / kGemmBK);

    f32x4 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 ra[Cfg::kLA], rb[Cfg::kLB];
    auto fetch = [&](int s) {
        const long long kb = k_lo + (long long)s * kGemmBK;
#pragma unroll
        for (int q = 0; q < Cfg::kLA; ++q) {
            const int idx = tid + q * kGemmThreads;
            const int row = idx / (Cfg::TM / 4), col = (idx % (Cfg::TM / 4)) * 4;
            const bool ok = idx < Cfg::kA4 && kb + row < k_hi;
            ra[q] = ok ? *(const float4 *)(A + (kb + row) * lda + i0 + col) : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < Cfg::kLB; ++q) {
            const int idx = tid + q * kGemmThreads;
            const int row = idx / (Cfg::TN / 4), col = (idx % (Cfg::TN / 4)) * 4;
            const bool ok = idx < Cfg::kB4 && kb + row < k_hi;
            const float *src = (B2 && j0 + col >= nb) ? B2 + (kb + row) * ldb2 + (j0 + col - nb) : B + (kb + row) * ldb + j0 + col;
            rb[q] = ok ? *(const float4 *)src : float4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stash = [&](int buf) {
        float *la = lds + buf * Cfg::kStageFloats, *lb = la + kGemmBK * Cfg::SA;
#pragma unroll
        for (int q = 0; q < Cfg::kLA; ++q) {
            const int idx = tid + q * kGemmThreads;
            if (idx < Cfg::kA4) *(float4 *)(la + (idx / (Cfg::TM / 4)) * Cfg::SA + (idx % (Cfg::TM / 4)) * 4) = ra[q];
        }
#pragma unroll
        for (int q = 0; q < Cfg::kLB; ++q) {
            const int idx = tid + q * kGemmThreads;
            if (idx < Cfg::kB4) *(float4 *)(lb + (idx / (Cfg::TN / 4)) * Cfg::SB + (idx % (Cfg::TN / 4)) * 4) = rb[q];
        }
    };

    if (stages > 0) {
        fetch(0);
        stash(0);
    }
    __syncthreads();
    for (int s = 0; s < stages; ++s) {
        const int buf = s & 1;
        if (s + 1 < stages) fetch(s + 1);  // global loads in flight across this stage's MFMAs
        const float *la = lds + buf * Cfg::kStageFloats + wr * MI * 16 + c;
        const float *lb = lds + buf * Cfg::kStageFloats + kGemmBK * Cfg::SA + wc * NI * 16 + c;
#pragma unroll
        for (int kk = 0; kk < kGemmBK / 4; ++kk) {
            float fa[MI], fb[NI];
#pragma unroll
            for (int a = 0; a < MI; ++a) fa[a] = la[(4 * kk + g) * Cfg::SA + 16 * a];
#pragma unroll
            for (int b = 0; b < NI; ++b) fb[b] = lb[(4 * kk + g) * Cfg::SB + 16 * b];
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int b = 0; b < NI; ++b) acc[a][b] = mfma16(fa[a], fb[b], acc[a][b]);
        }
        if (s + 1 < stages) stash(buf ^ 1);
        __syncthreads();
    }

    float *out = partial + (size_t)split * mo * no;
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(size_t)(i0 + (wr * MI + a) * 16 + 4 * g + r) * no + j0 + (wc * NI + b) * 16 + c] = acc[a][b][r];
}

// pass 2: 64 float4 column groups x 4 split quarters per workgroup; each thread sums its quarter of the splits (4 loads in
// flight), the quarters are combined through LDS in a fixed order -> deterministic, and latency- rather than issue-bound.
__device__ __forceinline__ void gemm_tn_reduce_block(const float *__restrict__ partial, int splits, int mo, int no, float *__restrict__ C,
                                                     long long ldc, float *__restrict__ C2, long long ldc2, int nb, int block,
                                                     double (*sh)[64][4]) {   // C2 != null: columns >= nb go to C2[i][col - nb]
    const int col4 = threadIdx.x & 63, quarter = threadIdx.x >> 6;
    const int idx4 = block * 64 + col4;  // float4 index into the mo x no output
    const int total4 = mo * no / 4;
    const size_t stride4 = (size_t)mo * no / 4;
    const float4 *p4 = reinterpret_cast<const float4 *>(partial);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (idx4 < total4) {
        const int per = (splits + 3) / 4;
        const int lo = quarter * per, hi = lo + per < splits ? lo + per : splits;
        int q = lo;
        for (; q + 4 <= hi; q += 4) {
            const float4 a = p4[(size_t)q * stride4 + idx4], b = p4[(size_t)(q + 1) * stride4 + idx4];
            const float4 c = p4[(size_t)(q + 2) * stride4 + idx4], d = p4[(size_t)(q + 3) * stride4 + idx4];
            s0 += ((double)a.x + (double)b.x) + ((double)c.x + (double)d.x);
            s1 += ((double)a.y + (double)b.y) + ((double)c.y + (double)d.y);
            s2 += ((double)a.z + (double)b.z) + ((double)c.z + (double)d.z);
            s3 += ((double)a.w + (double)b.w) + ((double)c.w + (double)d.w);
        }
        for (; q < hi; ++q) {
            const float4 a = p4[(size_t)q * stride4 + idx4];
            s0 += a.x, s1 += a.y, s2 += a.z, s3 += a.w;
        }
    }
    if (quarter > 0) {
        sh[quarter - 1][col4][0] = s0, sh[quarter - 1][col4][1] = s1, sh[quarter - 1][col4][2] = s2, sh[quarter - 1][col4][3] = s3;
    }
    __syncthreads();
    if (quarter == 0 && idx4 < total4) {
#pragma unroll
        for (int k = 0; k < 3; ++k) s0 += sh[k][col4][0], s1 += sh[k][col4][1], s2 += sh[k][col4][2], s3 += sh[k][col4][3];
        const int e = idx4 * 4, i = e / no, j = e % no;  // no % 4 == 0: the four values stay in one row
        float *out = (C2 && j >= nb) ? C2 + (long long)i * ldc2 + (j - nb) : C + (long long)i * ldc + j;
        out[0] = (float)s0, out[1] = (float)s1, out[2] = (float)s2, out[3] = (float)s3;
    }
}
// pass 2: 64 float4 column groups x 4 split quarters per workgroup; each thread sums its quarter of the splits (4 loads in
// flight), the quarters are combined through LDS in a fixed order -> deterministic, and latency- rather than issue-bound.
__global__ void __launch_bounds__(256) gemm_tn_reduce_kernel(const float *__restrict__ partial, int splits, int mo, int no,
                                                            float *__restrict__ C, long long ldc, float *__restrict__ C2 = nullptr,
                                                            long long ldc2 = 0, int nb = 0) {
    __shared__ double sh[3][64][4];
    gemm_tn_reduce_block(partial, splits, mo, no, C, ldc, C2, ldc2, nb, (int)blockIdx.x, sh);
}

// Column sums of per-workgroup partials [groups][cols] -> out0[col] (col < nb) / out1[col - nb]: one block per 64 columns, 4 group
// quarters combined through LDS in a fixed order (the recurrent update's bias gradients, csrc/lstm_seq.hip lstm_bias_final_kernel's sum).
__device__ __forceinline__ void colsum_reduce_block(const float *__restrict__ partial, int groups, int cols, int nb, float *__restrict__ out0,
                                                    float *__restrict__ out1, int block, float (*shf)[64]) {
    const int cl = threadIdx.x & 63, quarter = threadIdx.x >> 6;
    const int col = block * 64 + cl;
    const int per = (groups + 3) / 4;
    const int lo = quarter * per, hi = lo + per < groups ? lo + per : groups;
    float s0 = 0.0f, s1 = 0.0f;
    if (col < cols) {
        int i = lo;
        for (; i + 2 <= hi; i += 2) {
            s0 += partial[(size_t)i * cols + col];
            s1 += partial[(size_t)(i + 1) * cols + col];
        }
        if (i < hi) s0 += partial[(size_t)i * cols + col];
    }
    shf[quarter][cl] = s0 + s1;
    __syncthreads();
    if (quarter == 0 && col < cols) {
        const float s = (shf[0][cl] + shf[1][cl]) + (shf[2][cl] + shf[3][cl]);
        if (col < nb) out0[col] = s;
        else out1[col - nb] = s;
    }
}

// Several of these reductions in ONE launch (the recurrent update: three weight-gradient products + the bias column sums used to be
// four latency-bound launches of 10-14 us each in a row): block b belongs to the job whose block range holds it.
constexpr int kReduceJobsMax = 6;
struct ReduceJobs {
    pfa_reduce_job job[kReduceJobsMax];
    int first_block[kReduceJobsMax + 1];
    int n;
};
__global__ void __launch_bounds__(256) reduce_multi_kernel(ReduceJobs J) {
    __shared__ double sh[3][64][4];
    __shared__ float shf[4][64];
    int j = 0;
#pragma unroll
    for (int q = 1; q < kReduceJobsMax; ++q)
        if (q < J.n && (int)blockIdx.x >= J.first_block[q]) j = q;
    const pfa_reduce_job &jb = J.job[j];
    const int block = (int)blockIdx.x - J.first_block[j];
    if (jb.kind == 0)
        gemm_tn_reduce_block(jb.partial, jb.splits, jb.mo, jb.no, jb.c, jb.ldc, jb.c2, jb.ldc2, jb.nb, block, sh);
    else
        colsum_reduce_block(jb.partial, jb.splits, jb.no, jb.nb, jb.c, jb.c2, block, shf);
}

struct GemmTnPlan {
    int cfg;  // 0: 128x128, 1: 128x64, 2: 16x128, 3: 128x32, 4: 128x16, 5: 128x160, 6: 128x256 (the two-operand form)
    int tm, tn, tiles, splits;
    long long k_per_split;
};

static bool gemm_tn_plan(int mo, int no, long long K, GemmTnPlan &p) {
    if (mo % 128 == 0 && no % 128 == 0) {
        p.cfg = 0, p.tm = 128, p.tn = 128;
    } else if (mo % 128 == 0 && no % 64 == 0) {
        p.cfg = 1, p.tm = 128, p.tn = 64;
    } else if (mo % 128 == 0 && no % 160 == 0) {   // the encoder gradient of 160-float rows (configs[2]): one 32 x 160 strip per wave, 20 MFMAs
        p.cfg = 5, p.tm = 128, p.tn = 160;          // per 12 fragment reads (the 128 x 32 tile it used before: 4 per 5, LDS-issue bound)
    } else if (mo % 128 == 0 && no % 32 == 0) {
        p.cfg = 3, p.tm = 128, p.tn = 32;
    } else if (mo % 128 == 0 && no % 16 == 0) {
        p.cfg = 4, p.tm = 128, p.tn = 16;
    } else if (mo % 16 == 0 && no % 128 == 0) {
        p.cfg = 2, p.tm = 16, p.tn = 128;
    } else {
        return false;
    }
    p.tiles = (mo / p.tm) * (no / p.tn);
    // >= 2 workgroups per CU when k allows it, every split at least 8 slabs deep, split size a multiple of the slab
    long long want = (512 + p.tiles - 1) / p.tiles;
    const long long max_splits = (K + 8 * kGemmBK - 1) / (8 * kGemmBK);
    if (want > max_splits) want = max_splits;
    if (want < 1) want = 1;
    long long per = (K + want - 1) / want;
    per = (per + kGemmBK - 1) / kGemmBK * kGemmBK;
    p.k_per_split = per;
    p.splits = (int)((K + per - 1) / per);
    return true;
}

}  // namespace pfa

using namespace pfa;

extern "C" size_t pfa_gemm_tn_workspace_bytes(int32_t mo, int32_t no, int64_t k) {
    GemmTnPlan p;
    if (mo <= 0 || no <= 0 || k <= 0 || !gemm_tn_plan(mo, no, k, p)) return 0;
    return (size_t)p.splits * mo * no * sizeof(float);
}

static int gemm_tn_launch_partial(const float *a, int64_t lda, const float *b, int64_t ldb, int32_t mo, int32_t no, int64_t k, void *workspace,
                                  GemmTnPlan &p, hipStream_t st) {
    PFA_REQUIRE(a && b && workspace && mo > 0 && no > 0 && k > 0, "gemm_tn: bad arguments");
    PFA_REQUIRE(lda >= mo && ldb >= no && lda % 4 == 0 && ldb % 4 == 0, "gemm_tn: row strides must cover the tile and be 16-byte multiples");
    PFA_REQUIRE(((uintptr_t)a | (uintptr_t)b) % 16 == 0, "gemm_tn: operands must be 16-byte aligned");
    PFA_REQUIRE(gemm_tn_plan(mo, no, k, p), "gemm_tn: supported shapes are (128a x 16b) and (16a x 128b)");
    const dim3 grid((unsigned)p.tiles, (unsigned)p.splits);
    float *partial = (float *)workspace;
    if (p.cfg == 0)
        hipLaunchKernelGGL((gemm_tn_partial_kernel<2, 2, 4, 4>), grid, dim3(kGemmThreads), 0, st, a, (long long)lda, b, (long long)ldb,
                           mo, no, (long long)k, p.k_per_split, partial);
    else if (p.cfg == 1)
        hipLaunchKernelGGL((gemm_tn_partial_kernel<2, 2, 4, 2>), grid, dim3(kGemmThreads), 0, st, a, (long long)lda, b, (long long)ldb,
                           mo, no, (long long)k, p.k_per_split, partial);
    else if (p.cfg == 3)
        hipLaunchKernelGGL((gemm_tn_partial_kernel<2, 2, 4, 1>), grid, dim3(kGemmThreads), 0, st, a, (long long)lda, b, (long long)ldb,
                           mo, no, (long long)k, p.k_per_split, partial);
    else if (p.cfg == 5)
        hipLaunchKernelGGL((gemm_tn_partial_kernel<4, 1, 2, 10>), grid, dim3(kGemmThreads), 0, st, a, (long long)lda, b, (long long)ldb,
                           mo, no, (long long)k, p.k_per_split, partial);
    else if (p.cfg == 4)
        hipLaunchKernelGGL((gemm_tn_partial_kernel<4, 1, 2, 1>), grid, dim3(kGemmThreads), 0, st, a, (long long)lda, b, (long long)ldb,
                           mo, no, (long long)k, p.k_per_split, partial);
    else
        hipLaunchKernelGGL((gemm_tn_partial_kernel<1, 4, 1, 2>), grid, dim3(kGemmThreads), 0, st, a, (long long)lda, b, (long long)ldb,
                           mo, no, (long long)k, p.k_per_split, partial);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_gemm_tn_f32(const float *a, int64_t lda, const float *b, int64_t ldb, float *c, int64_t ldc, int32_t mo,
                               int32_t no, int64_t k, void *workspace, pfa_stream_t stream) {
    PFA_REQUIRE(c && ldc >= no, "gemm_tn: bad output");
    GemmTnPlan p;
    hipStream_t st = (hipStream_t)stream;
    ScopedKernelTimer timer("gemm_tn", st);
    if (int rc = gemm_tn_launch_partial(a, lda, b, ldb, mo, no, k, workspace, p, st)) return rc;
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)((mo * no / 4 + 63) / 64)), dim3(256), 0, st, (const float *)workspace, p.splits, mo, no, c,
                       (long long)ldc);
    PFA_LAUNCH_CHECK();
    return 0;
}

// The split partials only: `job` receives the reduction that finishes the product (kind 0; fill in c / ldc), for pfa_reduce_multi.
extern "C" int pfa_gemm_tn_partial_f32(const float *a, int64_t lda, const float *b, int64_t ldb, int32_t mo, int32_t no, int64_t k, void *workspace,
                                       pfa_reduce_job *job, pfa_stream_t stream) {
    PFA_REQUIRE(job != nullptr, "gemm_tn_partial: null job");
    GemmTnPlan p;
    hipStream_t st = (hipStream_t)stream;
    ScopedKernelTimer timer("gemm_tn", st);
    if (int rc = gemm_tn_launch_partial(a, lda, b, ldb, mo, no, k, workspace, p, st)) return rc;
    *job = pfa_reduce_job{0, p.splits, mo, no, (const float *)workspace, nullptr, 0, nullptr, 0, 0, 0};
    return 0;
}

// C0 = A^T B0 and C1 = A^T B1 in ONE pass over A (n0 = n1 = 128 columns each, mo a multiple of 128): the recurrent layer's two
// weight gradients dW_ih = dG^T xe and dW_hh = dG^T h_prev share their 512-column operand dG, which as two products was staged
// through LDS twice per k slab and fetched from HBM twice (268 MB per minibatch each time at BASELINE configs[2]).  One 128 x 256
// output tile per workgroup: 12 fragment reads per 32 MFMAs instead of 8 per 16.  Same split / fixed-order f64 reduction as above.
extern "C" size_t pfa_gemm_tn2_workspace_bytes(int32_t mo, int64_t k) {
    if (mo <= 0 || mo % 128 || k <= 0) return 0;
    GemmTnPlan p;
    if (!gemm_tn_plan(mo, 128, k, p)) return 0;     // (the same split count as the 128 x 128 tiling: tiles = mo / 128)
    return (size_t)p.splits * mo * 256 * sizeof(float);
}
static int gemm_tn2_launch_partial(const float *a, int64_t lda, const float *b0, int64_t ldb0, const float *b1, int64_t ldb1, int32_t mo, int64_t k,
                                   void *workspace, int *splits_out, hipStream_t st) {
    PFA_REQUIRE(a && b0 && b1 && workspace && mo > 0 && mo % 128 == 0 && k > 0, "gemm_tn2: bad arguments (mo must be a multiple of 128)");
    PFA_REQUIRE(lda >= mo && ldb0 >= 128 && ldb1 >= 128 && lda % 4 == 0 && ldb0 % 4 == 0 && ldb1 % 4 == 0,
                "gemm_tn2: row strides must cover the tile and be 16-byte multiples");
    PFA_REQUIRE(((uintptr_t)a | (uintptr_t)b0 | (uintptr_t)b1) % 16 == 0, "gemm_tn2: operands must be 16-byte aligned");
    const int tiles = mo / 128, no = 256;
    long long want = (512 + tiles - 1) / tiles;
    const long long max_splits = (k + 8 * kGemmBK - 1) / (8 * kGemmBK);
    if (want > max_splits) want = max_splits;
    if (want < 1) want = 1;
    long long per = (k + want - 1) / want;
    per = (per + kGemmBK - 1) / kGemmBK * kGemmBK;
    const int splits = (int)((k + per - 1) / per);
    PFA_REQUIRE((size_t)splits * mo * no * sizeof(float) <= pfa_gemm_tn2_workspace_bytes(mo, k), "gemm_tn2: split plan exceeds the workspace");
    hipLaunchKernelGGL((gemm_tn_partial_kernel<2, 2, 4, 8>), dim3((unsigned)tiles, (unsigned)splits), dim3(kGemmThreads), 0, st, a, (long long)lda, b0,
                       (long long)ldb0, mo, no, (long long)k, per, (float *)workspace, b1, (long long)ldb1, 128);
    PFA_LAUNCH_CHECK();
    *splits_out = splits;
    return 0;
}
extern "C" int pfa_gemm_tn2_f32(const float *a, int64_t lda, const float *b0, int64_t ldb0, const float *b1, int64_t ldb1, float *c0, int64_t ldc0,
                                float *c1, int64_t ldc1, int32_t mo, int64_t k, void *workspace, pfa_stream_t stream) {
    PFA_REQUIRE(c0 && c1 && ldc0 >= 128 && ldc1 >= 128, "gemm_tn2: bad outputs");
    hipStream_t st = (hipStream_t)stream;
    ScopedKernelTimer timer("gemm_tn", st);
    int splits = 0;
    if (int rc = gemm_tn2_launch_partial(a, lda, b0, ldb0, b1, ldb1, mo, k, workspace, &splits, st)) return rc;
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)((mo * 256 / 4 + 63) / 64)), dim3(256), 0, st, (const float *)workspace, splits, mo, 256, c0,
                       (long long)ldc0, c1, (long long)ldc1, 128);
    PFA_LAUNCH_CHECK();
    return 0;
}
extern "C" int pfa_gemm_tn2_partial_f32(const float *a, int64_t lda, const float *b0, int64_t ldb0, const float *b1, int64_t ldb1, int32_t mo, int64_t k,
                                        void *workspace, pfa_reduce_job *job, pfa_stream_t stream) {
    PFA_REQUIRE(job != nullptr, "gemm_tn2_partial: null job");
    hipStream_t st = (hipStream_t)stream;
    ScopedKernelTimer timer("gemm_tn", st);
    int splits = 0;
    if (int rc = gemm_tn2_launch_partial(a, lda, b0, ldb0, b1, ldb1, mo, k, workspace, &splits, st)) return rc;
    *job = pfa_reduce_job{0, splits, mo, 256, (const float *)workspace, nullptr, 0, nullptr, 0, 128, 0};
    return 0;
}

// Up to six reductions in ONE launch.  kind 0: the f64 fixed-order sum of `splits` partial products [mo][no] -> c[i][j] (c2 != NULL:
// columns j >= nb -> c2[i][j - nb]); kind 1: column sums of [splits][no] per-workgroup partials -> c[col] for col < nb, c2[col - nb] else.
extern "C" int pfa_reduce_multi(const pfa_reduce_job *jobs, int32_t njobs, pfa_stream_t stream) {
    PFA_REQUIRE(jobs && njobs >= 1 && njobs <= kReduceJobsMax, "reduce_multi: 1..%d jobs", kReduceJobsMax);
    ReduceJobs J{};
    J.n = njobs;
    int blocks = 0;
    for (int q = 0; q < njobs; ++q) {
        const pfa_reduce_job &jb = jobs[q];
        PFA_REQUIRE(jb.partial && jb.c && jb.splits >= 1 && jb.no >= 1 && (jb.kind == 1 || (jb.kind == 0 && jb.mo >= 1 && jb.no % 4 == 0 && jb.ldc >= (jb.c2 ? jb.nb : jb.no))),
                    "reduce_multi: bad job %d", q);
        PFA_REQUIRE(jb.kind == 0 || jb.nb >= jb.no || jb.c2, "reduce_multi: job %d needs c2 for columns >= nb", q);
        J.job[q] = jb;
        J.first_block[q] = blocks;
        blocks += jb.kind == 0 ? (jb.mo * jb.no / 4 + 63) / 64 : (jb.no + 63) / 64;
    }
    J.first_block[njobs] = blocks;
    ScopedKernelTimer timer("reduce_multi", (hipStream_t)stream);
    hipLaunchKernelGGL(reduce_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, J);
    PFA_LAUNCH_CHECK();
    return 0;
}
