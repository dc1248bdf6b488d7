// igemm.hip — fp32-MFMA implicit-GEMM kernels for the NatureCNN policy of BASELINE configs[3]
// (pufferlib/models.py:113-157 `Convolutional`: Conv2d(4,32,8,s4) - ReLU - Conv2d(32,64,4,s2) - ReLU - Conv2d(64,64,3,s1) - ReLU -
// Flatten - Linear(3136,512) - ReLU, heads Linear(512,A) / Linear(512,1); observations uint8 (4,84,84), `.float() / 255.0`).
//
// Every product of the forward pass and of loss.backward() is one of two contractions over v_mfma_f32_16x16x4_f32 (exact fp32
// FMA chains; the 1e-5 parity target rules out bf16):
//   rows form   C[m][n]  = sum_k A(m, k) * B[k][n]            forward conv / linear (m = output pixel or sample, k = patch index)
//                                                             and dX (m = INPUT pixel, k = (ky, kx, oc), A = gathered dOut)
//   weight form G[k][n]  = sum_m A(m, k) * D[m][n]            dW (contraction over the rows of the minibatch, split over row chunks)
// A(m, k) is never materialised: the tile loader computes the address of the patch element (implicit im2col / col2im), so a conv
// layer reads its input once per k-slab from L2/HBM and nothing else.  Activations are NHWC f32 (the patch index runs
// (ky, kx, ic) with ic contiguous: 16-float runs per row), the first layer reads the uint8 NCHW frames directly ((ic, ky, kx) with
// kx contiguous: 4-byte runs) and applies `/ 255.0` on the way in.  Weights are re-packed into the [k][n] matrices these orders
// need after every optimizer step (pfa_cnn_pack); gradients leave in torch's parameter layout.
//
// Tiling (both forms): 256 threads = 4 waves, k-slabs of 16 double-buffered in LDS k-major ([k][tile width], row stride == 16
// mod 32 floats so that the MFMA fragment reads — lane (c, g) reads [4 kk + g][16 a + c] — are conflict-free), global loads of
// slab s+1 in flight under the MFMAs of slab s.  rows form: 128 x (32 | 64) output tile, wave w owns rows 32w..32w+31.
// weight form: (64 | 128) x (32 | 64) tile of G per workgroup and a chunk of rows, partial sums reduced in f64 (deterministic).
#include "common.hpp"
#include "mlp_tile.hpp"

namespace pfa {

constexpr int kIgThreads = 256;
constexpr int kIgBK = 16;

enum IgAMode : int { kADense = 0, kAIm2colF32 = 1, kAIm2colU8 = 2, kACol2im = 3 };
enum IgEpilogue : int { kEpiNone = 0, kEpiBias = 1, kEpiBiasRelu = 2, kEpiMask = 3 };

// One conv layer (valid padding): input [N][IH][IW][IC] (NHWC f32) or uint8 [N][IC][IH][IW]; output [N][OH][OW][OC] NHWC.
struct IgGeom {
    int IC, IH, IW, OC, OH, OW, KH, KW, S;
};

struct IgA {
    int mode;
    const void *ptr;   // dense: float [M][lda]; im2col f32: NHWC input; im2col u8: NCHW frames; col2im: dOut NHWC [N][OH][OW][OC]
    long long lda;
    IgGeom g;
};

// Per-row part of the address (computed once per tile row): returns the element offset of patch element k = 0 and a validity
// descriptor the k part needs.
struct IgRow {
    long long base;   // element offset (dense: m*lda; im2col: (n, oy*S, ox*S) corner; col2im: n*OH*OW*OC)
    int y, x;         // col2im: input pixel coordinates
    bool ok;
};

__device__ __forceinline__ IgRow ig_row(const IgA &a, long long m, long long M) {
    IgRow r;
    r.ok = m < M;
    r.y = r.x = 0;
    r.base = 0;
    if (!r.ok) return r;
    if (a.mode == kADense) {
        r.base = m * a.lda;
    } else if (a.mode == kAIm2colF32) {
        const int ohw = a.g.OH * a.g.OW;
        const long long n = m / ohw;
        const int rem = (int)(m - n * ohw), oy = rem / a.g.OW, ox = rem - oy * a.g.OW;
        r.base = ((n * a.g.IH + (long long)oy * a.g.S) * a.g.IW + (long long)ox * a.g.S) * a.g.IC;
    } else if (a.mode == kAIm2colU8) {
        const int ohw = a.g.OH * a.g.OW;
        const long long n = m / ohw;
        const int rem = (int)(m - n * ohw), oy = rem / a.g.OW, ox = rem - oy * a.g.OW;
        r.base = (n * a.g.IC * a.g.IH + (long long)oy * a.g.S) * a.g.IW + (long long)ox * a.g.S;
    } else {   // col2im: m = input pixel (n, y, x)
        const int ihw = a.g.IH * a.g.IW;
        const long long n = m / ihw;
        const int rem = (int)(m - n * ihw);
        r.y = rem / a.g.IW;
        r.x = rem - r.y * a.g.IW;
        r.base = n * a.g.OH * a.g.OW * a.g.OC;
    }
    return r;
}

// Four consecutive patch elements k .. k+3 (k % 4 == 0) of a row.
__device__ __forceinline__ float4 ig_load4(const IgA &a, const IgRow &r, int k) {
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!r.ok) return z;
    if (a.mode == kADense) {
        return *reinterpret_cast<const float4 *>((const float *)a.ptr + r.base + k);
    } else if (a.mode == kAIm2colF32) {   // k = (ky*KW + kx)*IC + ic, IC % 4 == 0
        const int pix = k / a.g.IC, ic = k - pix * a.g.IC, ky = pix / a.g.KW, kx = pix - ky * a.g.KW;
        return *reinterpret_cast<const float4 *>((const float *)a.ptr + r.base + ((long long)ky * a.g.IW + kx) * a.g.IC + ic);
    } else if (a.mode == kAIm2colU8) {    // k = (ic*KH + ky)*KW + kx, KW % 4 == 0; observations.float() / 255.0 (models.py:150)
        const int kx = k % a.g.KW, t = k / a.g.KW, ky = t % a.g.KH, ic = t / a.g.KH;
        const uint8_t *p = (const uint8_t *)a.ptr + r.base + ((long long)ic * a.g.IH + ky) * a.g.IW + kx;
        return make_float4((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, (float)p[3] / 255.0f);
    } else {                              // k = (ky*KW + kx)*OC + oc: dOut[n][(y-ky)/S][(x-kx)/S][oc] where that is a pixel
        const int pix = k / a.g.OC, oc = k - pix * a.g.OC, ky = pix / a.g.KW, kx = pix - ky * a.g.KW;
        const int dy = r.y - ky, dx = r.x - kx;
        if (dy < 0 || dx < 0) return z;
        const int oy = dy / a.g.S, ox = dx / a.g.S;
        if (oy * a.g.S != dy || ox * a.g.S != dx || oy >= a.g.OH || ox >= a.g.OW) return z;
        return *reinterpret_cast<const float4 *>((const float *)a.ptr + r.base + ((long long)oy * a.g.OW + ox) * a.g.OC + oc);
    }
}

// ------------------------------------------------------------------------------------------------------------------ rows form
// C[m][n] = epilogue(sum_k A(m, k) B[k][n]); B row-major [K][ldb]; C row-major [M][ldc].  grid = (ceil(M/128), N / (16 NI)).
template <int NI>
__global__ void __launch_bounds__(kIgThreads, 2) igemm_rows_kernel(IgA A, long long M, int K, const float *__restrict__ B, int ldb,
                                                                  float *__restrict__ Cout, int ldc, int epi, const float *__restrict__ bias,
                                                                  const float *__restrict__ mask, int ldmask) {
    constexpr int TM = 128, TN = 16 * NI, SA = TM + 16, SB = TN % 32 == 0 ? TN + 16 : TN + 32;
    constexpr int kStage = kIgBK * (SA + SB);
    __shared__ float lds[2 * kStage];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, g = lane >> 4;
    const long long m0 = (long long)blockIdx.x * TM;
    const int n0 = blockIdx.y * TN;
    // A slab: 128 rows x 16 k: thread -> (row = idx / 4, quad = idx % 4), two rows per thread
    IgRow rows[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) rows[q] = ig_row(A, m0 + (tid + q * kIgThreads) / 4, M);
    const int kq = (tid & 3) * 4;
    f32x4 acc[2][NI];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int stages = (K + kIgBK - 1) / kIgBK;
    float4 ra[2], rb;
    constexpr int kB4 = kIgBK * TN / 4;   // float4 of a B slab (<= 256)
    auto fetch = [&](int s) {
        const int k0 = s * kIgBK;
#pragma unroll
        for (int q = 0; q < 2; ++q) ra[q] = k0 + kq < K ? ig_load4(A, rows[q], k0 + kq) : make_float4(0.f, 0.f, 0.f, 0.f);
        rb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < kB4) {
            const int row = tid / (TN / 4), col = (tid % (TN / 4)) * 4;
            if (k0 + row < K) rb = *reinterpret_cast<const float4 *>(B + (size_t)(k0 + row) * ldb + n0 + col);
        }
    };
    auto stash = [&](int buf) {
        float *la = lds + buf * kStage, *lb = la + kIgBK * SA;
#pragma unroll
        for (int q = 0; q < 2; ++q) {   // transpose into k-major: la[k][m]
            const int m = (tid + q * kIgThreads) / 4;
            la[(kq + 0) * SA + m] = ra[q].x;
            la[(kq + 1) * SA + m] = ra[q].y;
            la[(kq + 2) * SA + m] = ra[q].z;
            la[(kq + 3) * SA + m] = ra[q].w;
        }
        if (tid < kB4) *reinterpret_cast<float4 *>(lb + (tid / (TN / 4)) * SB + (tid % (TN / 4)) * 4) = rb;
    };
    fetch(0);
    stash(0);
    __syncthreads();
    for (int s = 0; s < stages; ++s) {
        const int buf = s & 1;
        if (s + 1 < stages) fetch(s + 1);
        const float *la = lds + buf * kStage + wv * 32 + c;
        const float *lb = lds + buf * kStage + kIgBK * SA + c;
#pragma unroll
        for (int kk = 0; kk < kIgBK / 4; ++kk) {
            float fa[2], fb[NI];
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[a] = la[(4 * kk + g) * SA + 16 * a];
#pragma unroll
            for (int b = 0; b < NI; ++b) fb[b] = lb[(4 * kk + g) * SB + 16 * b];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < NI; ++b) acc[a][b] = mfma16(fa[a], fb[b], acc[a][b]);
        }
        if (s + 1 < stages) stash(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long m = m0 + wv * 32 + a * 16 + 4 * g + r;
            if (m >= M) continue;
#pragma unroll
            for (int b = 0; b < NI; ++b) {
                const int n = n0 + b * 16 + c;
                float v = acc[a][b][r];
                if (epi == kEpiBias || epi == kEpiBiasRelu) v += bias[n];
                if (epi == kEpiBiasRelu) v = fmaxf(v, 0.0f);
                if (epi == kEpiMask) v = mask[(size_t)m * ldmask + n] > 0.0f ? v : 0.0f;   // relu' of the layer input, read where it was produced
                Cout[(size_t)m * ldc + n] = v;
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------- weight form
// partial[split][k][n] = sum over the split's rows of A(m, k) D[m][n]; grid = ((K / (16 MI*WR)) * (N / (16 NI*WC)), splits).
template <int WR, int WC, int MI, int NI>
__global__ void __launch_bounds__(kIgThreads, 2) igemm_weights_kernel(IgA A, long long M, int K, const float *__restrict__ D, int ldd, int N,
                                                                     long long rows_per_split, float *__restrict__ partial) {
    constexpr int TK = WR * MI * 16, TN = WC * NI * 16, SA = TK % 32 == 0 ? TK + 16 : TK + 32, SB = TN % 32 == 0 ? TN + 16 : TN + 32;
    constexpr int kStage = kIgBK * (SA + SB);
    constexpr int kA4 = kIgBK * TK / 4, kB4 = kIgBK * TN / 4;
    constexpr int kLA = (kA4 + kIgThreads - 1) / kIgThreads, kLB = (kB4 + kIgThreads - 1) / kIgThreads;
    __shared__ float lds[2 * kStage];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, g = lane >> 4;
    const int wr = wv / WC, wc = wv % WC;
    const int tiles_n = N / TN;
    const int k0 = (blockIdx.x / tiles_n) * TK, n0 = (blockIdx.x % tiles_n) * TN;
    const long long m_lo = (long long)blockIdx.y * rows_per_split;
    const long long m_hi = m_lo + rows_per_split < M ? m_lo + rows_per_split : M;
    const int stages = (int)((m_hi - m_lo + kIgBK - 1) / kIgBK);
    f32x4 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 ra[kLA], rb[kLB];
    auto fetch = [&](int s) {
        const long long mb = m_lo + (long long)s * kIgBK;
#pragma unroll
        for (int q = 0; q < kLA; ++q) {
            const int idx = tid + q * kIgThreads;
            const int row = idx / (TK / 4), col = (idx % (TK / 4)) * 4;
            ra[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < kA4 && mb + row < m_hi && k0 + col < K) ra[q] = ig_load4(A, ig_row(A, mb + row, M), k0 + col);
        }
#pragma unroll
        for (int q = 0; q < kLB; ++q) {
            const int idx = tid + q * kIgThreads;
            const int row = idx / (TN / 4), col = (idx % (TN / 4)) * 4;
            rb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < kB4 && mb + row < m_hi) rb[q] = *reinterpret_cast<const float4 *>(D + (size_t)(mb + row) * ldd + n0 + col);
        }
    };
    auto stash = [&](int buf) {
        float *la = lds + buf * kStage, *lb = la + kIgBK * SA;
#pragma unroll
        for (int q = 0; q < kLA; ++q) {
            const int idx = tid + q * kIgThreads;
            if (idx < kA4) *reinterpret_cast<float4 *>(la + (idx / (TK / 4)) * SA + (idx % (TK / 4)) * 4) = ra[q];
        }
#pragma unroll
        for (int q = 0; q < kLB; ++q) {
            const int idx = tid + q * kIgThreads;
            if (idx < kB4) *reinterpret_cast<float4 *>(lb + (idx / (TN / 4)) * SB + (idx % (TN / 4)) * 4) = rb[q];
        }
    };
    if (stages > 0) {
        fetch(0);
        stash(0);
    }
    __syncthreads();
    for (int s = 0; s < stages; ++s) {
        const int buf = s & 1;
        if (s + 1 < stages) fetch(s + 1);
        const float *la = lds + buf * kStage + wr * MI * 16 + c;
        const float *lb = lds + buf * kStage + kIgBK * SA + wc * NI * 16 + c;
#pragma unroll
        for (int kk = 0; kk < kIgBK / 4; ++kk) {
            float fa[MI], fb[NI];
#pragma unroll
            for (int a = 0; a < MI; ++a) fa[a] = la[(4 * kk + g) * SA + 16 * a];
#pragma unroll
            for (int b = 0; b < NI; ++b) fb[b] = lb[(4 * kk + g) * SB + 16 * b];
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int b = 0; b < NI; ++b) acc[a][b] = mfma16(fa[a], fb[b], acc[a][b]);
        }
        if (s + 1 < stages) stash(buf ^ 1);
        __syncthreads();
    }
    float *out = partial + (size_t)blockIdx.y * K * N;
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + (wr * MI + a) * 16 + 4 * g + r;
                if (k < K) out[(size_t)k * N + n0 + (wc * NI + b) * 16 + c] = acc[a][b][r];
            }
}

// Sum of the split partials (f64, fixed order) scattered into the gradient in torch's layout.  perm: 0 = out[k*N + n] (a
// transposed Linear: [K][N] kept), 1 = Linear weight [n][k], 2 = conv weight [oc = n][ic][ky][kx] from k = (ky*KW + kx)*IC + ic,
// 3 = conv weight from k = (ic*KH + ky)*KW + kx (the uint8 first layer).  accumulate: += (micro-batches of one minibatch).
__global__ void __launch_bounds__(256) igemm_weights_reduce_kernel(const float *__restrict__ partial, int splits, int K, int N, float *__restrict__ out,
                                                                  int perm, IgGeom g, int accumulate) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)K * N) return;
    double s = 0.0;
    for (int q = 0; q < splits; ++q) s += (double)partial[(size_t)q * K * N + i];
    const int k = (int)(i / N), n = (int)(i - (long long)k * N);
    long long p;
    if (perm == 0) p = i;
    else if (perm == 1) p = (long long)n * K + k;
    else if (perm == 2) {
        const int pix = k / g.IC, ic = k - pix * g.IC, ky = pix / g.KW, kx = pix - ky * g.KW;
        p = (((long long)n * g.IC + ic) * g.KH + ky) * g.KW + kx;
    } else if (perm == 3) {
        p = (long long)n * K + k;   // (ic, ky, kx) IS torch's order within an output channel
    } else {   // 4: Linear behind nn.Flatten of an NCHW tensor whose activations we keep NHWC: k = (y*IW + x)*IC + c -> c*IH*IW + y*IW + x
        const int pix = k / g.IC, cch = k - pix * g.IC;
        p = (long long)n * K + (long long)cch * g.IH * g.IW + pix;
    }
    out[p] = (accumulate ? out[p] : 0.0f) + (float)s;
}

// Column sums (bias gradients): out[n] (+)= sum_m D[m][n]; two deterministic stages.
constexpr int kIgColChunks = 256;
__global__ void __launch_bounds__(256) ig_colsum_partial_kernel(const float *__restrict__ D, long long M, int N, int ldd, double *__restrict__ partial) {
    const long long per = (M + kIgColChunks - 1) / kIgColChunks;
    const long long lo = (long long)blockIdx.y * per, hi = lo + per < M ? lo + per : M;
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    __shared__ double sh[4][64];
    double s = 0.0;
    if (n < N)
        for (long long m = lo + sl; m < hi; m += 4) s += (double)D[(size_t)m * ldd + n];
    sh[sl][threadIdx.x & 63] = s;
    __syncthreads();
    if (sl == 0 && n < N) partial[(size_t)blockIdx.y * N + n] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}
__global__ void __launch_bounds__(256) ig_colsum_final_kernel(const double *__restrict__ partial, int N, float *__restrict__ out, int accumulate) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    double s = 0.0;
    for (int q = 0; q < kIgColChunks; ++q) s += partial[(size_t)q * N + n];
    out[n] = (accumulate ? out[n] : 0.0f) + (float)s;
}

// Weight re-packing after an optimizer step: conv [OC][IC][KH][KW] -> forward B [k][OC] in the patch order of the layer's loader,
// and dX B [(ky*KW + kx)*OC + oc][IC]; Linear [N][K] -> [K][N] (forward) — its own layout serves dX.
__global__ void __launch_bounds__(256) ig_pack_conv_kernel(const float *__restrict__ w, IgGeom g, int u8_order, float *__restrict__ fwd,
                                                          float *__restrict__ dx) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)g.OC * g.IC * g.KH * g.KW;
    if (i >= total) return;
    const int kx = (int)(i % g.KW);
    long long t = i / g.KW;
    const int ky = (int)(t % g.KH);
    t /= g.KH;
    const int ic = (int)(t % g.IC), oc = (int)(t / g.IC);
    const float v = w[i];
    const long long kf = u8_order ? ((long long)ic * g.KH + ky) * g.KW + kx : ((long long)ky * g.KW + kx) * g.IC + ic;
    fwd[kf * g.OC + oc] = v;
    if (dx) dx[(((long long)ky * g.KW + kx) * g.OC + oc) * g.IC + ic] = v;
}
// Linear(C*H*W, N) behind nn.Flatten (models.py:133): torch's column index is NCHW (c*H*W + y*W + x), our activation rows are NHWC
// ((y*W + x)*C + c).  perm_out [N][K'] (columns re-ordered; B of dX) and t_out [K'][N] (B of the forward).
__global__ void __launch_bounds__(256) ig_pack_fc_kernel(const float *__restrict__ w, int N, int Cc, int HW, float *__restrict__ perm_out,
                                                        float *__restrict__ t_out) {
    const int K = Cc * HW;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * K) return;
    const int kp = (int)(i % K), n = (int)(i / K);
    const int pix = kp / Cc, cch = kp - pix * Cc;
    const float v = w[(size_t)n * K + (size_t)cch * HW + pix];
    perm_out[i] = v;
    t_out[(size_t)kp * N + n] = v;
}
__global__ void __launch_bounds__(256) ig_transpose_kernel(const float *__restrict__ w, int N, int K, float *__restrict__ out) {   // [N][K] -> [K][N]
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * K) return;
    const int k = (int)(i % K), n = (int)(i / K);
    out[(size_t)k * N + n] = w[i];
}

static int ig_check_a(const pfa_igemm_operand *a, int K) {
    PFA_REQUIRE(a && a->ptr, "igemm: null A operand");
    PFA_REQUIRE(a->mode >= 0 && a->mode <= 3, "igemm: bad A mode %d", a->mode);
    PFA_REQUIRE(K % 4 == 0, "igemm: K must be a multiple of 4 (got %d)", K);
    if (a->mode == kADense) PFA_REQUIRE(a->lda % 4 == 0 && a->lda >= K, "igemm: dense lda must be a multiple of 4 and >= K");
    if (a->mode == kAIm2colF32) PFA_REQUIRE(a->IC % 4 == 0 && K == a->KH * a->KW * a->IC, "igemm: im2col f32 needs IC %% 4 == 0 and K == KH*KW*IC");
    if (a->mode == kAIm2colU8) PFA_REQUIRE(a->KW % 4 == 0 && K == a->KH * a->KW * a->IC, "igemm: im2col u8 needs KW %% 4 == 0 and K == IC*KH*KW");
    if (a->mode == kACol2im) PFA_REQUIRE(a->OC % 4 == 0 && K == a->KH * a->KW * a->OC, "igemm: col2im needs OC %% 4 == 0 and K == KH*KW*OC");
    return 0;
}
static IgA ig_make_a(const pfa_igemm_operand *a) {
    IgA r;
    r.mode = a->mode;
    r.ptr = a->ptr;
    r.lda = a->lda;
    r.g = IgGeom{a->IC, a->IH, a->IW, a->OC, a->OH, a->OW, a->KH, a->KW, a->S};
    return r;
}

}  // namespace pfa

using namespace pfa;

extern "C" int pfa_igemm_rows(const pfa_igemm_operand *a, int64_t M, int32_t K, const float *B, int32_t ldb, int32_t N, float *C, int32_t ldc,
                              int32_t epilogue, const float *bias, const float *mask, int32_t ldmask, pfa_stream_t stream) {
    if (int rc = ig_check_a(a, K)) return rc;
    PFA_REQUIRE(M >= 0 && B && C && N >= 16 && N % 16 == 0 && ldb >= N && ldb % 4 == 0 && ldc >= N, "igemm.rows: bad shapes (N must be a multiple of 16)");
    PFA_REQUIRE(epilogue >= 0 && epilogue <= 3, "igemm.rows: bad epilogue");
    PFA_REQUIRE((epilogue != kEpiBias && epilogue != kEpiBiasRelu) || bias, "igemm.rows: bias epilogue without a bias vector");
    PFA_REQUIRE(epilogue != kEpiMask || (mask && ldmask >= N), "igemm.rows: mask epilogue without a mask");
    if (M == 0) return 0;
    const IgA A = ig_make_a(a);
    const unsigned gm = (unsigned)((M + 127) / 128);
    ScopedKernelTimer timer("igemm_rows", (hipStream_t)stream);
    if (N % 64 == 0)
        hipLaunchKernelGGL(igemm_rows_kernel<4>, dim3(gm, N / 64), dim3(kIgThreads), 0, (hipStream_t)stream, A, (long long)M, (int)K, B, (int)ldb, C,
                           (int)ldc, (int)epilogue, bias, mask, (int)ldmask);
    else if (N % 32 == 0)
        hipLaunchKernelGGL(igemm_rows_kernel<2>, dim3(gm, N / 32), dim3(kIgThreads), 0, (hipStream_t)stream, A, (long long)M, (int)K, B, (int)ldb, C,
                           (int)ldc, (int)epilogue, bias, mask, (int)ldmask);
    else
        hipLaunchKernelGGL(igemm_rows_kernel<1>, dim3(gm, N / 16), dim3(kIgThreads), 0, (hipStream_t)stream, A, (long long)M, (int)K, B, (int)ldb, C,
                           (int)ldc, (int)epilogue, bias, mask, (int)ldmask);
    PFA_LAUNCH_CHECK();
    return 0;
}

static int ig_splits(int64_t M, int tiles) {
    int64_t s = (M + 2047) / 2048;          // >= 2048 rows per split
    const int64_t cap = 2048 / (tiles > 0 ? tiles : 1) + 1;
    if (s > cap) s = cap;
    if (s > 512) s = 512;
    return (int)(s < 1 ? 1 : s);
}

extern "C" size_t pfa_igemm_weights_workspace_bytes(int64_t M, int32_t K, int32_t N) {
    const int tiles = ((K + 63) / 64) * ((N + 31) / 32);
    return (size_t)ig_splits(M, tiles) * (size_t)K * (size_t)N * sizeof(float) + 256;
}

extern "C" int pfa_igemm_weights(const pfa_igemm_operand *a, int64_t M, int32_t K, const float *D, int32_t ldd, int32_t N, float *out,
                                 int32_t perm, int32_t accumulate, void *workspace, pfa_stream_t stream) {
    if (int rc = ig_check_a(a, K)) return rc;
    PFA_REQUIRE(M >= 1 && D && out && workspace && N >= 16 && N % 16 == 0 && ldd >= N && ldd % 4 == 0, "igemm.weights: bad shapes");
    PFA_REQUIRE(perm >= 0 && perm <= 4, "igemm.weights: bad permutation");
    const IgA A = ig_make_a(a);
    float *partial = (float *)workspace;
    int splits;
    long long rps;
    ScopedKernelTimer timer("igemm_weights", (hipStream_t)stream);
    if (N % 64 == 0) {          // 128 x 64 tiles: WR=4 (MI=2) x WC=1 (NI=4)
        const int tiles = ((K + 127) / 128) * (N / 64);
        splits = ig_splits(M, tiles);
        rps = (((M + splits - 1) / splits) + 15) / 16 * 16;
        hipLaunchKernelGGL((igemm_weights_kernel<4, 1, 2, 4>), dim3(tiles, splits), dim3(kIgThreads), 0, (hipStream_t)stream, A, (long long)M, (int)K, D,
                           (int)ldd, (int)N, rps, partial);
    } else if (N % 32 == 0) {   // 128 x 32
        const int tiles = ((K + 127) / 128) * (N / 32);
        splits = ig_splits(M, tiles);
        rps = (((M + splits - 1) / splits) + 15) / 16 * 16;
        hipLaunchKernelGGL((igemm_weights_kernel<4, 1, 2, 2>), dim3(tiles, splits), dim3(kIgThreads), 0, (hipStream_t)stream, A, (long long)M, (int)K, D,
                           (int)ldd, (int)N, rps, partial);
    } else {                    // 128 x 16
        const int tiles = ((K + 127) / 128) * (N / 16);
        splits = ig_splits(M, tiles);
        rps = (((M + splits - 1) / splits) + 15) / 16 * 16;
        hipLaunchKernelGGL((igemm_weights_kernel<4, 1, 2, 1>), dim3(tiles, splits), dim3(kIgThreads), 0, (hipStream_t)stream, A, (long long)M, (int)K, D,
                           (int)ldd, (int)N, rps, partial);
    }
    PFA_LAUNCH_CHECK();
    const long long total = (long long)K * N;
    hipLaunchKernelGGL(igemm_weights_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partial, splits, (int)K,
                       (int)N, out, (int)perm, A.g, (int)accumulate);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t pfa_colsum_workspace_bytes(int32_t N) { return (size_t)kIgColChunks * (size_t)N * sizeof(double); }

extern "C" int pfa_colsum(const float *D, int64_t M, int32_t N, int32_t ldd, float *out, int32_t accumulate, void *workspace, pfa_stream_t stream) {
    PFA_REQUIRE(D && out && workspace && M >= 1 && N >= 1 && ldd >= N, "colsum: bad arguments");
    hipLaunchKernelGGL(ig_colsum_partial_kernel, dim3((N + 63) / 64, kIgColChunks), dim3(256), 0, (hipStream_t)stream, D, (long long)M, (int)N, (int)ldd,
                       (double *)workspace);
    PFA_LAUNCH_CHECK();
    hipLaunchKernelGGL(ig_colsum_final_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const double *)workspace, (int)N, out,
                       (int)accumulate);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_cnn_pack_conv(const float *w, const pfa_igemm_operand *geom, int32_t u8_order, float *fwd, float *dx, pfa_stream_t stream) {
    PFA_REQUIRE(w && geom && fwd, "cnn.pack_conv: null buffer");
    const IgA A = ig_make_a(geom);
    const long long total = (long long)A.g.OC * A.g.IC * A.g.KH * A.g.KW;
    hipLaunchKernelGGL(ig_pack_conv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, A.g, (int)u8_order, fwd, dx);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_cnn_transpose(const float *w, int32_t N, int32_t K, float *out, pfa_stream_t stream) {
    PFA_REQUIRE(w && out && N >= 1 && K >= 1, "cnn.transpose: bad arguments");
    const long long total = (long long)N * K;
    hipLaunchKernelGGL(ig_transpose_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (int)N, (int)K, out);
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_cnn_pack_fc(const float *w, int32_t N, int32_t channels, int32_t hw, float *perm_out, float *t_out, pfa_stream_t stream) {
    PFA_REQUIRE(w && perm_out && t_out && N >= 1 && channels >= 1 && hw >= 1, "cnn.pack_fc: bad arguments");
    const long long total = (long long)N * channels * hw;
    hipLaunchKernelGGL(ig_pack_fc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (int)N, (int)channels, (int)hw,
                       perm_out, t_out);
    PFA_LAUNCH_CHECK();
    return 0;
}
