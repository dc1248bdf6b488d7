// igemm.hip — fp32-MFMA implicit-GEMM kernels for the NatureCNN policy of BASELINE configs[3]
// (pufferlib/models.py:113-157 `Convolutional`: Conv2d(4,32,8,s4) - ReLU - Conv2d(32,64,4,s2) - ReLU - Conv2d(64,64,3,s1) - ReLU -
// Flatten - Linear(3136,512) - ReLU, heads Linear(512,A) / Linear(512,1); observations uint8 (4,84,84), `.float() / 255.0`).
//
// Every product of the forward pass and of loss.backward() is one of two contractions over v_mfma_f32_16x16x4_f32 (exact fp32
// FMA chains; the 1e-5 parity target rules out bf16):
//   rows form   C[m][n]  = sum_k A(m, k) * B[n][k]            forward conv / linear (m = output pixel or sample, k = patch index)
//                                                             and dX (m = INPUT pixel, k = (tap, oc), A = gathered dOut)
//   weight form G[k][n]  = sum_m A(m, k) * D[m][n]            dW (+ column sums of D = the bias gradient), split over row chunks
// A(m, k) is never materialised: the tile loader computes the address of the patch element (implicit im2col / col2im), so a conv
// layer reads its input once per k-slab from L2/HBM and nothing else.  Activations are NHWC f32 (patch index (ky, kx, ic), ic
// contiguous); the first layer reads the uint8 NCHW frames directly ((ic, ky, kx), kx contiguous: one aligned 32-bit word = four
// patch elements) as floats 0..255 — the `/ 255.0` is folded into its packed weights (forward) and into the reduction of its
// weight gradient, one VALU conversion per element instead of four.  dX of a stride-S layer runs as S*S phases (input pixels with equal
// (y mod S, x mod S) share the (KH/S)(KW/S) kernel taps that reach them), so no multiply is spent on the structural zeros of a
// transposed strided convolution.
//
// Address arithmetic is the cost that competes with the MFMAs (VALU and MFMA issue from the same SIMD port and do not overlap),
// so it is incremental: a thread keeps the (tap, channel) decomposition of its fixed k-quad (rows form) or its fixed row pair's
// (n, oy, ox) (weight form) and advances it by one slab with adds and compares; divisions happen once per tile.
//
// Tiling: 256 threads = 4 waves, k-slabs of 16, double-buffered in LDS, global loads of slab s+1 in flight under the MFMAs of
// slab s.  rows form: (64 MI) x (16 NI) output tile, both operands row-major in LDS with a 20-float row stride; lane (c, g) reads
// ONE float4 per 16 x 16 fragment and slab (row c, columns 4g..4g+3 — the MFMA's k-slot g then stands for k = 4g + kk in step kk,
// for A and B alike), 2 lanes per bank.  weight form: (64 KJ) x (16 NI) tile of G per workgroup (KJ = 1..4: the height that pads K
// least) and a chunk of rows; the partial sums
// of the row splits are reduced in f64 in a fixed order (deterministic) and scattered into torch's parameter layout.
#include "common.hpp"
#include "mlp_tile.hpp"

#ifndef PFA_IG_TILE64
#define PFA_IG_TILE64 0
#endif

namespace pfa {

constexpr int kIgThreads = 256;
constexpr int kIgBK = 16;
constexpr int kIgLS = kIgBK + 4;   // LDS row stride (floats) of a [row][16 k] slab

enum IgAMode : int { kADense = 0, kAIm2colF32 = 1, kAIm2colU8 = 2, kACol2im = 3 };
enum IgEpilogue : int { kEpiNone = 0, kEpiBias = 1, kEpiBiasRelu = 2, kEpiMask = 3 };

// One conv layer (valid padding): input [N][IH][IW][IC] (NHWC f32) or uint8 [N][IC][IH][IW]; output [N][OH][OW][OC] NHWC.
struct IgGeom {
    int IC, IH, IW, OC, OH, OW, KH, KW, S;
};

struct IgA {
    int mode;
    const void *ptr;   // dense: float [M][lda]; im2col f32: NHWC input; im2col u8: NCHW frames; col2im: dOut NHWC [N][OH][OW][OC]
    int lda;
    IgGeom g;
    int JH, JW, HP, WP;   // col2im: taps per phase (KH/S, KW/S) and pixels per phase and frame (IH/S, IW/S)
};

// -------------------------------------------------------------------------------------------------- the k part of an address
// Where patch elements k .. k+3 sit relative to a row's base element, kept incrementally (k advances by one slab at a time).
struct IgK {
    int off;          // element offset to add to the row base
    int p, q, ch;     // running decomposition: im2col f32 (ky, kx, ic) / u8 (ic, ky, kx) / col2im (jy, jx, oc)
};

template <int MODE>
__device__ __forceinline__ IgK ig_k_init(const IgA &a, int k) {
    IgK s;
    if (MODE == kADense) {
        s.p = s.q = s.ch = 0;
        s.off = k;
    } else if (MODE == kAIm2colF32) {   // k = (ky*KW + kx)*IC + ic
        const int pix = k / a.g.IC;
        s.ch = k - pix * a.g.IC;
        s.p = pix / a.g.KW;
        s.q = pix - s.p * a.g.KW;
        s.off = (s.p * a.g.IW + s.q) * a.g.IC + s.ch;
    } else if (MODE == kAIm2colU8) {    // k = (ic*KH + ky)*KW + kx
        s.ch = k % a.g.KW;
        const int t = k / a.g.KW;
        s.q = t % a.g.KH;
        s.p = t / a.g.KH;
        s.off = (s.p * a.g.IH + s.q) * a.g.IW + s.ch;
    } else {                              // k = (jy*JW + jx)*OC + oc; the row base is dOut pixel (yy, xx): tap (jy, jx) reads (yy-jy, xx-jx)
        const int pix = k / a.g.OC;
        s.ch = k - pix * a.g.OC;
        s.p = pix / a.JW;
        s.q = pix - s.p * a.JW;
        s.off = -(s.p * a.g.OW + s.q) * a.g.OC + s.ch;
    }
    return s;
}

template <int MODE>
__device__ __forceinline__ void ig_k_advance(const IgA &a, IgK &s) {   // k += 16
    if (MODE == kADense) {
        s.off += kIgBK;
    } else if (MODE == kAIm2colF32) {
        s.ch += kIgBK;
        while (s.ch >= a.g.IC) {
            s.ch -= a.g.IC;
            if (++s.q == a.g.KW) {
                s.q = 0;
                ++s.p;
            }
        }
        s.off = (s.p * a.g.IW + s.q) * a.g.IC + s.ch;
    } else if (MODE == kAIm2colU8) {
        s.ch += kIgBK;
        while (s.ch >= a.g.KW) {
            s.ch -= a.g.KW;
            if (++s.q == a.g.KH) {
                s.q = 0;
                ++s.p;
            }
        }
        s.off = (s.p * a.g.IH + s.q) * a.g.IW + s.ch;
    } else {
        s.ch += kIgBK;
        while (s.ch >= a.g.OC) {
            s.ch -= a.g.OC;
            if (++s.q == a.JW) {
                s.q = 0;
                ++s.p;
            }
        }
        s.off = -(s.p * a.g.OW + s.q) * a.g.OC + s.ch;
    }
}

// ------------------------------------------------------------------------------------------------ the row part of an address
struct IgRow {
    int base;        // element offset of the row's first patch element (col2im: of dOut pixel (yy, xx), possibly outside the image)
    int y, x;        // col2im: (yy, xx); tap (jy, jx) reads dOut pixel (yy - jy, xx - jx) when that is inside the image
};

// rows form, col2im: m = (n, yy, xx) counts the input pixels of phase (py, px): y = yy*S + py, x = xx*S + px
template <int MODE>
__device__ __forceinline__ IgRow ig_row(const IgA &a, int m) {
    IgRow r;
    r.y = r.x = 0;
    if (MODE == kADense) {
        r.base = m * a.lda;
    } else if (MODE == kAIm2colF32 || MODE == kAIm2colU8) {
        const int ohw = a.g.OH * a.g.OW;
        const int n = m / ohw, rem = m - n * ohw, oy = rem / a.g.OW, ox = rem - oy * a.g.OW;
        r.base = MODE == kAIm2colF32 ? ((n * a.g.IH + oy * a.g.S) * a.g.IW + ox * a.g.S) * a.g.IC
                                       : (n * a.g.IC * a.g.IH + oy * a.g.S) * a.g.IW + ox * a.g.S;
    } else {
        const int hw = a.HP * a.WP;
        const int n = m / hw, rem = m - n * hw, yy = rem / a.WP, xx = rem - yy * a.WP;
        r.base = ((n * a.g.OH + yy) * a.g.OW + xx) * a.g.OC;
        r.y = yy;
        r.x = xx;
    }
    return r;
}

#ifdef PFA_IG_BOUNDS   // debug builds (tools/experiments/igemm_index_check.hip): record the first out-of-range access instead of making it
__device__ long long ig_dbg_limits[3];   // elements of A, B, C
__device__ int ig_dbg[8];                // [0] hits, [1] operand (0 A, 1 B, 2 C), [2] index lo, [3] index hi, [4] blockIdx.x, [5] threadIdx.x, [6] aux
__device__ __forceinline__ bool ig_dbg_bad(int operand, long long idx, long long span, int aux) {
    if (idx >= 0 && idx + span <= ig_dbg_limits[operand]) return false;
    if (atomicAdd(&ig_dbg[0], 1) == 0) {
        ig_dbg[1] = operand;
        ig_dbg[2] = (int)(idx & 0xFFFFFFFF);
        ig_dbg[3] = (int)(idx >> 32);
        ig_dbg[4] = blockIdx.x;
        ig_dbg[5] = threadIdx.x;
        ig_dbg[6] = aux;
    }
    return true;
}
#define IG_DBG_BAD(op, idx, span, aux) ig_dbg_bad(op, idx, span, aux)
#else
#define IG_DBG_BAD(op, idx, span, aux) false
#endif

template <int MODE>
__device__ __forceinline__ float4 ig_load4(const IgA &a, const IgRow &r, const IgK &k) {
    if (MODE == kAIm2colU8) {   // four bytes = one aligned word, as floats 0..255; the `/ 255.0` of models.py:150 rides in the other operand
        const uint32_t w = *reinterpret_cast<const uint32_t *>((const uint8_t *)a.ptr + (r.base + k.off));   // (packed weights / dW scale)
        return make_float4((float)(w & 0xFFu), (float)((w >> 8) & 0xFFu), (float)((w >> 16) & 0xFFu), (float)(w >> 24));
    }
    if (MODE == kACol2im && ((unsigned)(r.y - k.p) >= (unsigned)a.g.OH || (unsigned)(r.x - k.q) >= (unsigned)a.g.OW))
        return make_float4(0.f, 0.f, 0.f, 0.f);
    if (IG_DBG_BAD(0, (long long)r.base + k.off, 4, k.off)) return make_float4(0.f, 0.f, 0.f, 0.f);
    return *reinterpret_cast<const float4 *>((const float *)a.ptr + (r.base + k.off));
}

// ------------------------------------------------------------------------------------------------------------------ rows form
// C[row(m)][n] = epilogue(sum_k A(m, k) B[n][k]); B row-major [N][ldb] (k contiguous); C row-major, row stride ldc.
// grid = (ceil(M / (64 MI)), N / (16 NI), phases); col2im: blockIdx.z = phase, M = pixels per phase, B = B0 + phase * N * ldb.
template <int MODE, int MI, int NI>
__global__ void __launch_bounds__(kIgThreads, 2) igemm_rows_kernel(IgA A, int M, int K, const float *__restrict__ B, int ldb, int N,
                                                                  float *__restrict__ Cout, int ldc, int epi, const float *__restrict__ bias,
                                                                  const float *__restrict__ mask, int ldmask) {
    constexpr int TM = 64 * MI, TN = 16 * NI;
    constexpr int kStage = (TM + TN) * kIgLS;
    __shared__ __attribute__((aligned(16))) float lds[2 * kStage];
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id(), c = lane & 15, g = lane >> 4;
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2); neighbouring row tiles read overlapping patches of the
    // same frames, so the tile index is permuted to give every XCD one contiguous run of tiles (grid.x is a multiple of 8).
    const int per_xcd = gridDim.x >> 3;
    const int tile_m = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int m0 = tile_m * TM, n0 = blockIdx.y * TN;
    if (m0 >= M) return;
    const int py = MODE == kACol2im ? (int)blockIdx.z / A.g.S : 0, px = MODE == kACol2im ? (int)blockIdx.z % A.g.S : 0;
    B += (size_t)blockIdx.z * N * ldb;
    // loader role: slab row (tid >> 2) + 64 q, k-quad (tid & 3) * 4
    const int lr = tid >> 2, kq = (tid & 3) * 4;
    IgRow rows[MI];
#pragma unroll
    for (int q = 0; q < MI; ++q) {
        const int m = m0 + lr + 64 * q;
        rows[q] = ig_row<MODE>(A, m < M ? m : M - 1);      // rows past the end repeat the last one; the epilogue drops them
    }
    IgK kc = ig_k_init<MODE>(A, kq);
    const float *bp = B + (size_t)(n0 + (lr < TN ? lr : 0)) * ldb + kq;
    f32x4 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int stages = K / kIgBK;
    // global -> register prefetch one slab deep (two deep was measured: the extra registers cost occupancy, 60 vs 80 TFLOP/s),
    // register -> LDS double buffer
    float4 ra[MI], rb;
    auto fetch = [&](float4(&fa)[MI], float4 &fb) {   // the slab the cursor points at, then advance
#pragma unroll
        for (int q = 0; q < MI; ++q) fa[q] = ig_load4<MODE>(A, rows[q], kc);
        if (lr < TN && !IG_DBG_BAD(1, bp - B, 4, lr)) fb = *reinterpret_cast<const float4 *>(bp);
        ig_k_advance<MODE>(A, kc);
        bp += kIgBK;
    };
    auto stash = [&](int buf, const float4(&fa)[MI], const float4 &fb) {
        float *la = lds + buf * kStage, *lb = la + TM * kIgLS;
#pragma unroll
        for (int q = 0; q < MI; ++q) *reinterpret_cast<float4 *>(la + (lr + 64 * q) * kIgLS + kq) = fa[q];
        if (lr < TN) *reinterpret_cast<float4 *>(lb + lr * kIgLS + kq) = fb;
    };
    auto multiply = [&](int buf) {
        const float *la = lds + buf * kStage + (wv * 16 * MI + c) * kIgLS + 4 * g;
        const float *lb = lds + buf * kStage + TM * kIgLS + c * kIgLS + 4 * g;
        float4 fa[MI], fb[NI];
#pragma unroll
        for (int a = 0; a < MI; ++a) fa[a] = *reinterpret_cast<const float4 *>(la + a * 16 * kIgLS);
#pragma unroll
        for (int b = 0; b < NI; ++b) fb[b] = *reinterpret_cast<const float4 *>(lb + b * 16 * kIgLS);
        // step kk of the slab uses component kk of every fragment; consecutive MFMAs go to different accumulators
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b) acc[a][b] = mfma16(fa[a].x, fb[b].x, acc[a][b]);
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b) acc[a][b] = mfma16(fa[a].y, fb[b].y, acc[a][b]);
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b) acc[a][b] = mfma16(fa[a].z, fb[b].z, acc[a][b]);
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b) acc[a][b] = mfma16(fa[a].w, fb[b].w, acc[a][b]);
    };
    fetch(ra, rb);
    stash(0, ra, rb);
    __syncthreads();
    for (int s = 0; s < stages; ++s) {
        if (s + 1 < stages) fetch(ra, rb);
        multiply(s & 1);
        if (s + 1 < stages) stash((s & 1) ^ 1, ra, rb);
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < MI; ++a) {
        int en = 0, eyy = 0, exx = 0;
        if (MODE == kACol2im) {   // phase-local pixel of the first of this lane's four rows; the next ones by carry
            const int m = m0 + wv * 16 * MI + a * 16 + 4 * g, hw = A.HP * A.WP;
            en = m / hw;
            const int rem = m - en * hw;
            eyy = rem / A.WP;
            exx = rem - eyy * A.WP;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wv * 16 * MI + a * 16 + 4 * g + r;
            size_t orow = (size_t)m;
            if (MODE == kACol2im) {   // -> input pixel
                orow = ((size_t)en * A.g.IH + (size_t)(eyy * A.g.S + py)) * A.g.IW + (size_t)(exx * A.g.S + px);
                if (++exx == A.WP) {
                    exx = 0;
                    if (++eyy == A.HP) {
                        eyy = 0;
                        ++en;
                    }
                }
            }
            if (m >= M) continue;
#pragma unroll
            for (int b = 0; b < NI; ++b) {
                const int n = n0 + b * 16 + c;
                float v = acc[a][b][r];
                if (epi == kEpiBias || epi == kEpiBiasRelu) v += bias[n];
                if (epi == kEpiBiasRelu) v = fmaxf(v, 0.0f);
                if (epi == kEpiMask) v = mask[orow * ldmask + n] > 0.0f ? v : 0.0f;   // relu' of the layer input, read where it was produced
                if (IG_DBG_BAD(2, (long long)(orow * ldc + n), 1, m)) continue;
                Cout[orow * ldc + n] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------- rows form on the bf16 matrix path (opt-in)
// The same contraction with every fp32 operand split into three bf16 pieces in the loader (a = hi + mid + lo, round to nearest each
// time; 24 mantissa bits in all) and every 16 x 16 x 32 tile product issued as the six partial products above 2^-24 relative — lo.hi,
// hi.lo, mid.mid, mid.hi, hi.mid, hi.hi, small terms first — on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: 6/16 of the fp32
// MFMA's matrix time per product.  The uint8 first layer needs three (its A values are integers 0..255: one exact bf16 piece).
// Measured against f64 the result is as close as the fp32 fma chain (tools/experiments/bf16_split_accuracy.py, gemm_bf16x6.hip);
// it is NOT the bit pattern of the fp32 kernel, so it is off unless pfa_igemm_set_products(1) asks for it (bench.py --products
// bf16x6, tests/test_gpu_cnn.py).  32-deep slabs, three bf16 planes [row][32 k] of 80-byte rows in LDS, single-buffered (46 KB for
// the 128 x 64 tile: three workgroups per CU), the next slab's global loads in flight under the products.
typedef __bf16 ig_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ig_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ig_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t ig_pk_bf16(float a, float b) {   // two floats -> two bf16 (v_cvt_pk_bf16_f32, nearest even), a low
    ig_f32x2 v = {a, b};
    ig_bf16x2 r = __builtin_convertvector(v, ig_bf16x2);
    return *reinterpret_cast<uint32_t *>(&r);
}
__device__ __forceinline__ float ig_bf16_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float ig_bf16_hi(uint32_t p) { return __uint_as_float(p & 0xFFFF0000u); }
__device__ __forceinline__ void ig_split4(const float4 &x, uint2 &h, uint2 &m, uint2 &l) {
    h.x = ig_pk_bf16(x.x, x.y);
    h.y = ig_pk_bf16(x.z, x.w);
    const float r0 = x.x - ig_bf16_lo(h.x), r1 = x.y - ig_bf16_hi(h.x), r2 = x.z - ig_bf16_lo(h.y), r3 = x.w - ig_bf16_hi(h.y);
    m.x = ig_pk_bf16(r0, r1);
    m.y = ig_pk_bf16(r2, r3);
    l.x = ig_pk_bf16(r0 - ig_bf16_lo(m.x), r1 - ig_bf16_hi(m.x));
    l.y = ig_pk_bf16(r2 - ig_bf16_lo(m.y), r3 - ig_bf16_hi(m.y));
}

constexpr int kIgSplitBK = 32;
constexpr int kIgSplitRS = 80;   // bytes per LDS row of a plane: 32 bf16 + 16 bytes of padding (b128 fragment reads hit all banks once)

template <int MODE, int MI, int NI>
__global__ void __launch_bounds__(kIgThreads, 2) igemm_rows_split_kernel(IgA A, int M, int K, const float *__restrict__ B, int ldb, int N,
                                                                        float *__restrict__ Cout, int ldc, int epi, const float *__restrict__ bias,
                                                                        const float *__restrict__ mask, int ldmask) {
    constexpr int TM = 64 * MI, TN = 16 * NI, RS = kIgSplitRS, PLANE = (TM + TN) * RS;
    constexpr int RPP = 32, LA = TM / RPP, LB = TN / RPP;      // loader: 8 threads per slab row, 32 rows per pass
    constexpr bool kAInt = MODE == kAIm2colU8;                // A = bytes as floats: exact in ONE bf16 piece
    static_assert(TN % RPP == 0 && LB >= 1 && LB <= 2, "the split form takes 32- and 64-column tiles");
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id(), c = lane & 15, g = lane >> 4;
    const int per_xcd = gridDim.x >> 3;
    const int tile_m = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int m0 = tile_m * TM, n0 = blockIdx.y * TN;
    if (m0 >= M) return;
    const int py = MODE == kACol2im ? (int)blockIdx.z / A.g.S : 0, px = MODE == kACol2im ? (int)blockIdx.z % A.g.S : 0;
    B += (size_t)blockIdx.z * N * ldb;
    const int lr = tid >> 3, kq = (tid & 7) * 4;              // loader role: slab rows lr + 32 q, k-quad kq
    IgRow rows[LA];
#pragma unroll
    for (int q = 0; q < LA; ++q) {
        const int m = m0 + lr + RPP * q;
        rows[q] = ig_row<MODE>(A, m < M ? m : M - 1);
    }
    IgK kc = ig_k_init<MODE>(A, kq);
    const float *bp = B + (size_t)(n0 + lr) * ldb + kq;
    f32x4 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int stages = K / kIgSplitBK;
    float4 ra[LA], rb0, rb1 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&]() {   // the slab the cursor points at, then advance
#pragma unroll
        for (int q = 0; q < LA; ++q) ra[q] = ig_load4<MODE>(A, rows[q], kc);
        rb0 = *reinterpret_cast<const float4 *>(bp);
        if (LB > 1) rb1 = *reinterpret_cast<const float4 *>(bp + (size_t)RPP * ldb);
        ig_k_advance<MODE>(A, kc);
        ig_k_advance<MODE>(A, kc);
        bp += kIgSplitBK;
    };
    auto stash = [&]() {
        uint2 h, m, l;
#pragma unroll
        for (int q = 0; q < LA; ++q) {
            unsigned char *p = lds + (lr + RPP * q) * RS + kq * 2;
            if (kAInt) {
                *reinterpret_cast<uint2 *>(p) = make_uint2(ig_pk_bf16(ra[q].x, ra[q].y), ig_pk_bf16(ra[q].z, ra[q].w));
            } else {
                ig_split4(ra[q], h, m, l);
                *reinterpret_cast<uint2 *>(p) = h;
                *reinterpret_cast<uint2 *>(p + PLANE) = m;
                *reinterpret_cast<uint2 *>(p + 2 * PLANE) = l;
            }
        }
        unsigned char *pb = lds + (TM + lr) * RS + kq * 2;
        ig_split4(rb0, h, m, l);
        *reinterpret_cast<uint2 *>(pb) = h;
        *reinterpret_cast<uint2 *>(pb + PLANE) = m;
        *reinterpret_cast<uint2 *>(pb + 2 * PLANE) = l;
        if (LB > 1) {
            ig_split4(rb1, h, m, l);
            *reinterpret_cast<uint2 *>(pb + RPP * RS) = h;
            *reinterpret_cast<uint2 *>(pb + RPP * RS + PLANE) = m;
            *reinterpret_cast<uint2 *>(pb + RPP * RS + 2 * PLANE) = l;
        }
    };
    const unsigned char *la = lds + (wv * 16 * MI + c) * RS + g * 16, *lb = lds + (TM + c) * RS + g * 16;
    auto multiply = [&]() {
        ig_bf16x8 fa[kAInt ? 1 : 3][MI], fb[3][NI];
#pragma unroll
        for (int p = 0; p < (kAInt ? 1 : 3); ++p)
#pragma unroll
            for (int a = 0; a < MI; ++a) fa[p][a] = *reinterpret_cast<const ig_bf16x8 *>(la + p * PLANE + a * 16 * RS);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int b = 0; b < NI; ++b) fb[p][b] = *reinterpret_cast<const ig_bf16x8 *>(lb + p * PLANE + b * 16 * RS);
        // (piece of A, piece of B): 0 = hi, 1 = mid, 2 = lo; small terms first, consecutive MFMAs on different accumulators
        constexpr int kTerms = kAInt ? 3 : 6;
        constexpr int TA[6] = {kAInt ? 0 : 2, 0, kAInt ? 0 : 1, 1, 0, 0}, TB[6] = {kAInt ? 2 : 0, kAInt ? 1 : 2, kAInt ? 0 : 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < kTerms; ++t)
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int b = 0; b < NI; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[TA[t]][a], fb[TB[t]][b], acc[a][b], 0, 0, 0);
    };
    if (stages > 0) fetch();
    for (int s = 0; s < stages; ++s) {
        stash();
        __syncthreads();
        if (s + 1 < stages) fetch();          // the next slab's global loads fly under this slab's products
        multiply();
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < MI; ++a) {
        int en = 0, eyy = 0, exx = 0;
        if (MODE == kACol2im) {
            const int m = m0 + wv * 16 * MI + a * 16 + 4 * g, hw = A.HP * A.WP;
            en = m / hw;
            const int rem = m - en * hw;
            eyy = rem / A.WP;
            exx = rem - eyy * A.WP;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wv * 16 * MI + a * 16 + 4 * g + r;
            size_t orow = (size_t)m;
            if (MODE == kACol2im) {
                orow = ((size_t)en * A.g.IH + (size_t)(eyy * A.g.S + py)) * A.g.IW + (size_t)(exx * A.g.S + px);
                if (++exx == A.WP) {
                    exx = 0;
                    if (++eyy == A.HP) {
                        eyy = 0;
                        ++en;
                    }
                }
            }
            if (m >= M) continue;
#pragma unroll
            for (int b = 0; b < NI; ++b) {
                const int n = n0 + b * 16 + c;
                float v = acc[a][b][r];
                if (epi == kEpiBias || epi == kEpiBiasRelu) v += bias[n];
                if (epi == kEpiBiasRelu) v = fmaxf(v, 0.0f);
                if (epi == kEpiMask) v = mask[orow * ldmask + n] > 0.0f ? v : 0.0f;
                Cout[orow * ldc + n] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- weight form
// partial[split][k][n] = sum over the split's rows of A(m, k) D[m][n], and (k-tile 0 only) colpart[split][n] = sum of D[m][n];
// grid = ((K / 128 rounded up) * (N / (16 NI)), splits).  Wave w owns k rows 32w .. 32w+31 of the 128 x (16 NI) tile.
struct IgM {   // (n, oy, ox) of a row of an im2col operand, advanced by 16 rows per slab
    int n, oy, ox;
};
template <int MODE>
__device__ __forceinline__ int ig_m_base(const IgA &a, const IgM &r) {
    return MODE == kAIm2colF32 ? ((r.n * a.g.IH + r.oy * a.g.S) * a.g.IW + r.ox * a.g.S) * a.g.IC
                                 : (r.n * a.g.IC * a.g.IH + r.oy * a.g.S) * a.g.IW + r.ox * a.g.S;
}

template <int MODE, int KJ, int NI>
__global__ void __launch_bounds__(kIgThreads, 2) igemm_weights_kernel(IgA A, int M, int K, const float *__restrict__ D, int ldd, int N,
                                                                     int rows_per_split, float *__restrict__ partial, double *__restrict__ colpart) {
    constexpr int TK = 64 * KJ, TN = 16 * NI, SA = TK + 16, SB = TN % 32 == 0 ? TN + 16 : TN + 32;
    constexpr int kStage = kIgBK * (SA + SB);
    constexpr int kDThreads = kIgBK * TN / 4;
    __shared__ __attribute__((aligned(16))) float lds[2 * kStage];
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id(), c = lane & 15, g = lane >> 4;
    const int tiles_n = N / TN;
    const int ktile = blockIdx.x / tiles_n, k0 = ktile * TK, n0 = (blockIdx.x % tiles_n) * TN;
    const int m_lo = blockIdx.y * rows_per_split;
    const int m_hi = m_lo + rows_per_split < M ? m_lo + rows_per_split : M;
    const int stages = m_hi > m_lo ? (m_hi - m_lo + kIgBK - 1) / kIgBK : 0;
    // loader role: A row (tid >> 4) of the slab, k-quads (tid & 15) * 4 + 64 j, j < KJ — the k part is fixed for the whole kernel,
    // the row part is one cursor per thread.  Wave w owns k rows 16 KJ w .. 16 KJ (w + 1) - 1 of the tile (KJ fragments).
    const int ar = tid >> 4, aq = (tid & 15) * 4;
    bool k_ok[KJ];
    IgK kc[KJ];
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
        const int ak = k0 + aq + 64 * j;
        k_ok[j] = ak < K;
        kc[j] = ig_k_init<MODE>(A, k_ok[j] ? ak : 0);
    }
    IgM rm;
    {
        const int m = m_lo + ar;
        if (MODE == kADense) {
            rm.n = m;
            rm.oy = rm.ox = 0;
        } else {
            const int ohw = A.g.OH * A.g.OW;
            rm.n = m / ohw;
            const int rem = m - rm.n * ohw;
            rm.oy = rem / A.g.OW;
            rm.ox = rem - rm.oy * A.g.OW;
        }
    }
    const int dr = tid / (TN / 4), dc = (tid % (TN / 4)) * 4;
    f32x4 acc[KJ][NI];
#pragma unroll
    for (int a = 0; a < KJ; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    double colsum = 0.0;
    float4 ra[KJ], rb;
    int mrow = m_lo;          // first row of the slab the cursors point at
    auto fetch = [&](float4(&fa)[KJ], float4 &fb) {
        IgRow r;
        r.y = r.x = 0;
        r.base = MODE == kADense ? rm.n * A.lda : ig_m_base<MODE>(A, rm);
        const bool row_ok = mrow + ar < m_hi;
#pragma unroll
        for (int j = 0; j < KJ; ++j) fa[j] = (row_ok && k_ok[j]) ? ig_load4<MODE>(A, r, kc[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == kADense) {
            rm.n += kIgBK;
        } else {
            rm.ox += kIgBK;
            while (rm.ox >= A.g.OW) {
                rm.ox -= A.g.OW;
                if (++rm.oy == A.g.OH) {
                    rm.oy = 0;
                    ++rm.n;
                }
            }
        }
        fb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < kDThreads && mrow + dr < m_hi) fb = *reinterpret_cast<const float4 *>(D + (size_t)(mrow + dr) * ldd + n0 + dc);
        mrow += kIgBK;
    };
    auto stash = [&](int buf, const float4(&fa)[KJ], const float4 &fb) {
        float *la = lds + buf * kStage, *lb = la + kIgBK * SA;
#pragma unroll
        for (int j = 0; j < KJ; ++j) *reinterpret_cast<float4 *>(la + ar * SA + aq + 64 * j) = fa[j];
        if (tid < kDThreads) *reinterpret_cast<float4 *>(lb + dr * SB + dc) = fb;
    };
    auto multiply = [&](int buf) {
        const float *la = lds + buf * kStage + wv * 16 * KJ + c;
        const float *lb = lds + buf * kStage + kIgBK * SA + c;
#pragma unroll
        for (int kk = 0; kk < kIgBK / 4; ++kk) {
            float fa[KJ], fb[NI];
#pragma unroll
            for (int a = 0; a < KJ; ++a) fa[a] = la[(4 * kk + g) * SA + 16 * a];
#pragma unroll
            for (int b = 0; b < NI; ++b) fb[b] = lb[(4 * kk + g) * SB + 16 * b];
#pragma unroll
            for (int a = 0; a < KJ; ++a)
#pragma unroll
                for (int b = 0; b < NI; ++b) acc[a][b] = mfma16(fa[a], fb[b], acc[a][b]);
        }
        if (ktile == 0 && tid < TN) {   // bias gradient: column sums of this slab of D (rows past the split are zero in LDS)
            const float *lcol = lds + buf * kStage + kIgBK * SA + tid;
#pragma unroll
            for (int r = 0; r < kIgBK; ++r) colsum += (double)lcol[r * SB];
        }
    };
    if (stages > 0) {
        fetch(ra, rb);
        stash(0, ra, rb);
    }
    __syncthreads();
    for (int s = 0; s < stages; ++s) {
        if (s + 1 < stages) fetch(ra, rb);
        multiply(s & 1);
        if (s + 1 < stages) stash((s & 1) ^ 1, ra, rb);
        __syncthreads();
    }
    float *out = partial + (size_t)blockIdx.y * K * N;
#pragma unroll
    for (int a = 0; a < KJ; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + (wv * KJ + a) * 16 + 4 * g + r;
                if (k < K) out[(size_t)k * N + n0 + b * 16 + c] = acc[a][b][r];
            }
    if (ktile == 0 && tid < TN) colpart[(size_t)blockIdx.y * N + n0 + tid] = colsum;
}

// Sum of the split partials (f64, fixed order) scattered into the gradient in torch's layout.  perm: 0 = out[k*N + n] (a
// transposed Linear: [K][N] kept), 1 = Linear weight [n][k], 2 = conv weight [oc = n][ic][ky][kx] from k = (ky*KW + kx)*IC + ic,
// 3 = conv weight from k = (ic*KH + ky)*KW + kx, divided by 255 (the uint8 first layer: its loader hands out raw bytes).  Elements
// K*N .. K*N + N - 1: the bias gradient from the column-sum partials.  accumulate: += (micro-batches of one minibatch).
// 64 elements per workgroup, the splits dealt round-robin to its four waves and combined as (w0 + w1) + (w2 + w3).
__global__ void __launch_bounds__(256) igemm_weights_reduce_kernel(const float *__restrict__ partial, const double *__restrict__ colpart, int splits, int K,
                                                                  int N, float *__restrict__ out, float *__restrict__ bias_out, int perm, IgGeom g,
                                                                  int accumulate) {
    __shared__ double sh[4][64];
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + e;
    const long long kn = (long long)K * N;
    double s = 0.0;
    if (i < kn) {
        for (int q = sl; q < splits; q += 4) s += (double)partial[(size_t)q * kn + i];
    } else if (i < kn + N) {
        for (int q = sl; q < splits; q += 4) s += colpart[(size_t)q * N + (i - kn)];
    }
    sh[sl][e] = s;
    __syncthreads();
    if (sl != 0 || i >= kn + N) return;
    s = (sh[0][e] + sh[1][e]) + (sh[2][e] + sh[3][e]);
    if (i >= kn) {
        if (bias_out == nullptr) return;
        const int n = (int)(i - kn);
        bias_out[n] = (accumulate ? bias_out[n] : 0.0f) + (float)s;
        return;
    }
    const int k = (int)(i / N), n = (int)(i - (long long)k * N);
    long long p;
    if (perm == 0) p = i;
    else if (perm == 1) p = (long long)n * K + k;
    else if (perm == 2) {
        const int pix = k / g.IC, ic = k - pix * g.IC, ky = pix / g.KW, kx = pix - ky * g.KW;
        p = (((long long)n * g.IC + ic) * g.KH + ky) * g.KW + kx;
    } else if (perm == 3) {
        p = (long long)n * K + k;   // (ic, ky, kx) IS torch's order within an output channel
        s /= 255.0;
    } else {   // 4: Linear behind nn.Flatten of an NCHW tensor whose activations we keep NHWC: k = (y*IW + x)*IC + c -> c*IH*IW + y*IW + x
        const int pix = k / g.IC, cch = k - pix * g.IC;
        p = (long long)n * K + (long long)cch * g.IH * g.IW + pix;
    }
    out[p] = (accumulate ? out[p] : 0.0f) + (float)s;
}

// Column sums on their own (a bias gradient without a weight gradient next to it): out[n] (+)= sum_m D[m][n]; two deterministic stages.
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

// Weight re-packing after an optimizer step.  Conv [OC][IC][KH][KW] -> forward B [OC][k] with k in the loader's patch order
// (the uint8 first layer keeps torch's order and takes the / 255 of the observations), and dX B [phase][IC][(jy*JW + jx)*OC + oc] with
// phase = py*S + px and kernel tap (ky, kx) = (py + jy*S, px + jx*S).
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
    const int K = g.IC * g.KH * g.KW;
    if (fwd) {   // the uint8 loader hands out raw bytes: its weights carry the `/ 255.0` of models.py:150
        if (u8_order) fwd[(long long)oc * K + ((long long)ic * g.KH + ky) * g.KW + kx] = v / 255.0f;
        else fwd[(long long)oc * K + ((long long)ky * g.KW + kx) * g.IC + ic] = v;
    }
    if (dx) {
        const int JH = g.KH / g.S, JW = g.KW / g.S, py = ky % g.S, px = kx % g.S, jy = ky / g.S, jx = kx / g.S;
        const long long KP = (long long)JH * JW * g.OC;
        dx[((long long)(py * g.S + px) * g.IC + ic) * KP + ((long long)jy * JW + jx) * g.OC + oc] = v;
    }
}
// Linear(C*H*W, N) behind nn.Flatten (models.py:133): torch's column index is NCHW (c*H*W + y*W + x), our activation rows are NHWC
// ((y*W + x)*C + c).  perm_out [N][K'] (columns re-ordered: B of the forward) and t_out [K'][N] (B of dX).
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

static int ig_check_a(const pfa_igemm_operand *a, int64_t M, int K) {
    PFA_REQUIRE(a && a->ptr, "igemm: null A operand");
    PFA_REQUIRE(a->mode >= 0 && a->mode <= 3, "igemm: bad A mode %d", a->mode);
    PFA_REQUIRE(K >= 4 && K % 4 == 0, "igemm: K must be a multiple of 4 (got %d)", K);
    PFA_REQUIRE(((uintptr_t)a->ptr & 15) == 0, "igemm: the A operand must be 16-byte aligned");
    if (a->mode == kADense) {
        PFA_REQUIRE(a->lda % 4 == 0 && a->lda >= K, "igemm: dense lda must be a multiple of 4 and >= K");
        PFA_REQUIRE(M * a->lda < (1ll << 31), "igemm: operand too large for 32-bit element offsets (split the rows)");
        return 0;
    }
    PFA_REQUIRE(a->S >= 1 && a->KH >= 1 && a->KW >= 1 && a->OH == (a->IH - a->KH) / a->S + 1 && a->OW == (a->IW - a->KW) / a->S + 1,
                "igemm: inconsistent conv geometry");
    const int64_t frames = a->mode == kACol2im ? M / ((int64_t)a->IH * a->IW) + 1 : M / ((int64_t)a->OH * a->OW) + 1;
    PFA_REQUIRE(frames * a->IC * a->IH * a->IW < (1ll << 31) && frames * a->OC * a->OH * a->OW < (1ll << 31),
                "igemm: operand too large for 32-bit element offsets (split the rows)");
    if (a->mode == kAIm2colF32) PFA_REQUIRE(a->IC % 4 == 0 && K == a->KH * a->KW * a->IC, "igemm: im2col f32 needs IC %% 4 == 0 and K == KH*KW*IC");
    if (a->mode == kAIm2colU8)
        PFA_REQUIRE(a->KW % 4 == 0 && a->IW % 4 == 0 && a->S % 4 == 0 && (a->IC * a->IH * a->IW) % 4 == 0 && K == a->KH * a->KW * a->IC,
                    "igemm: im2col u8 needs KW, IW, S multiples of 4 (aligned 4-byte patch runs) and K == IC*KH*KW");
    if (a->mode == kACol2im)
        PFA_REQUIRE(a->OC % 4 == 0 && K == a->KH * a->KW * a->OC && a->KH % a->S == 0 && a->KW % a->S == 0 && a->IH % a->S == 0 && a->IW % a->S == 0 &&
                        true,
                    "igemm: col2im needs OC %% 4 == 0, K == KH*KW*OC, and KH, KW, IH, IW multiples of the stride");
    return 0;
}
static int g_ig_products = 0;   // 0: exact fp32 MFMA products (default); 1: six-term bf16 split products in the rows form

static IgA ig_make_a(const pfa_igemm_operand *a) {
    IgA r;
    r.mode = a->mode;
    r.ptr = a->ptr;
    r.lda = (int)a->lda;
    r.g = IgGeom{a->IC, a->IH, a->IW, a->OC, a->OH, a->OW, a->KH, a->KW, a->S};
    r.JH = r.JW = r.HP = r.WP = 0;
    if (a->mode == kACol2im) {
        r.JH = a->KH / a->S;
        r.JW = a->KW / a->S;
        r.HP = a->IH / a->S;
        r.WP = a->IW / a->S;
    }
    return r;
}

// Launch plan of the weight form, shared by the workspace query and the launch: tile width over n, number of (k, n) tiles, row
// splits (>= 2048 rows each, about 2048 workgroups in all, at most 1024) and rows per split (a multiple of the k-slab).
struct IgWeightPlan {
    int tn, kj, tiles, splits;   // tile = (64 kj) x tn
    int rows_per_split;
};
static IgWeightPlan ig_weight_plan(int64_t M, int K, int N) {
    IgWeightPlan p;
    p.tn = N % 64 == 0 ? 64 : N % 32 == 0 ? 32 : 16;
    // tile height over k: the one of 256 / 192 / 128 that pads K least (ties: the taller — more MFMAs per loaded D row and barrier)
    p.kj = 2;
    int best = (K + 127) / 128 * 128;
    for (int kj = 3; kj <= 4; ++kj) {
        const int padded = (K + 64 * kj - 1) / (64 * kj) * (64 * kj);
        if (padded <= best) {
            best = padded;
            p.kj = kj;
        }
    }
    if (K <= 64) p.kj = 1;      // (one 64-row k-tile: the encoder gradient of a 64-float observation row)
    p.tiles = ((K + 64 * p.kj - 1) / (64 * p.kj)) * (N / p.tn);
    int64_t s = (M + 2047) / 2048;
    const int64_t cap = 2048 / p.tiles + 1;
    if (s > cap) s = cap;
    if (s > 1024) s = 1024;
    // A small G (few tiles) over many rows would leave most CUs idle at >= 2048 rows per split: go down to 256 rows per split until
    // ~1024 workgroups exist, as long as the partial sums stay small (<= 16 MB written and re-read).
    if (s * p.tiles < 1024) {
        int64_t want = (1024 + p.tiles - 1) / p.tiles;
        const int64_t by_rows = (M + 255) / 256, by_bytes = (16ll << 20) / ((int64_t)K * N * 4);
        if (want > by_rows) want = by_rows;
        if (want > by_bytes) want = by_bytes;
        if (want > s) s = want;
    }
    p.splits = (int)(s < 1 ? 1 : s);
    p.rows_per_split = (int)((((M + p.splits - 1) / p.splits) + 15) / 16 * 16);
    return p;
}
static size_t ig_partial_bytes(const IgWeightPlan &p, int K, int N) { return align_up((size_t)p.splits * (size_t)K * (size_t)N * sizeof(float), 256); }

}  // namespace pfa

using namespace pfa;

extern "C" int pfa_igemm_rows(const pfa_igemm_operand *a, int64_t M, int32_t K, const float *B, int32_t ldb, int32_t N, float *C, int32_t ldc,
                              int32_t epilogue, const float *bias, const float *mask, int32_t ldmask, pfa_stream_t stream) {
    if (int rc = ig_check_a(a, M, K)) return rc;
    PFA_REQUIRE(M >= 0 && M < (1ll << 31) && B && C && N >= 16 && N % 16 == 0 && ldc >= N, "igemm.rows: bad shapes (N must be a multiple of 16)");
    PFA_REQUIRE(((uintptr_t)B & 15) == 0 && ldb % 4 == 0, "igemm.rows: B must be 16-byte aligned with ldb a multiple of 4");
    PFA_REQUIRE(epilogue >= 0 && epilogue <= 3, "igemm.rows: bad epilogue");
    PFA_REQUIRE((epilogue != kEpiBias && epilogue != kEpiBiasRelu) || bias, "igemm.rows: bias epilogue without a bias vector");
    PFA_REQUIRE(epilogue != kEpiMask || (mask && ldmask >= N), "igemm.rows: mask epilogue without a mask");
    if (M == 0) return 0;
    const IgA A = ig_make_a(a);
    int phases = 1, Kp = K;
    long long Mp = M;
    if (A.mode == kACol2im) {   // S*S phases of M / (S*S) pixels, each contracting over its own K / (S*S) taps x channels
        phases = A.g.S * A.g.S;
        PFA_REQUIRE(M % ((long long)A.g.IH * A.g.IW) == 0, "igemm.rows: col2im rows must be whole frames");
        Mp = M / phases;
        Kp = K / phases;
    }
    PFA_REQUIRE(Kp % kIgBK == 0 && ldb >= Kp, "igemm.rows: the contraction length (per phase) must be a multiple of 16 and ldb >= it");
    ScopedKernelTimer timer("igemm_rows", (hipStream_t)stream);
#define PFA_IG_ROWS(MODE, MI, NI)                                                                                                          \
    hipLaunchKernelGGL((igemm_rows_kernel<MODE, MI, NI>), dim3((unsigned)(((Mp + 64 * MI - 1) / (64 * MI) + 7) / 8 * 8), N / (16 * NI), phases), \
                       dim3(kIgThreads), 0, (hipStream_t)stream, A, (int)Mp, (int)Kp, B, (int)ldb, (int)N, C, (int)ldc, (int)epilogue, \
                       bias, mask, (int)ldmask)
#define PFA_IG_ROWS_MODE(MI, NI)                   \
    switch (A.mode) {                              \
        case kADense: PFA_IG_ROWS(kADense, MI, NI); break;           \
        case kAIm2colF32: PFA_IG_ROWS(kAIm2colF32, MI, NI); break;   \
        case kAIm2colU8: PFA_IG_ROWS(kAIm2colU8, MI, NI); break;     \
        default: PFA_IG_ROWS(kACol2im, MI, NI); break;               \
    }
    // A dense product over few rows (one rollout step of the width-general policies: 4096 rows) fills the chip only with 64-row tiles.
    const int tn = N % 64 == 0 ? 64 : N % 32 == 0 ? 32 : 16;
    const bool few = A.mode == kADense && ((Mp + (tn == 16 ? 255 : 127)) / (tn == 16 ? 256 : 128)) * (N / tn) < 256;
    if (g_ig_products == 1 && !few && tn >= 32 && Kp % kIgSplitBK == 0) {   // opt-in: the six-term bf16 form of the same products
#define PFA_IG_SPLIT(MODE, NI)                                                                                                             \
    hipLaunchKernelGGL((igemm_rows_split_kernel<MODE, 2, NI>), dim3((unsigned)(((Mp + 127) / 128 + 7) / 8 * 8), N / (16 * NI), phases),    \
                       dim3(kIgThreads), 0, (hipStream_t)stream, A, (int)Mp, (int)Kp, B, (int)ldb, (int)N, C, (int)ldc, (int)epilogue,   \
                       bias, mask, (int)ldmask)
#define PFA_IG_SPLIT_MODE(NI)                                        \
    switch (A.mode) {                                                \
        case kADense: PFA_IG_SPLIT(kADense, NI); break;              \
        case kAIm2colF32: PFA_IG_SPLIT(kAIm2colF32, NI); break;      \
        case kAIm2colU8: PFA_IG_SPLIT(kAIm2colU8, NI); break;        \
        default: PFA_IG_SPLIT(kACol2im, NI); break;                  \
    }
        if (tn == 64) {
            PFA_IG_SPLIT_MODE(4)
        } else {
            PFA_IG_SPLIT_MODE(2)
        }
#undef PFA_IG_SPLIT_MODE
#undef PFA_IG_SPLIT
    } else if (few) {
        if (tn == 64) {
            PFA_IG_ROWS(kADense, 1, 4);
        } else if (tn == 32) {
            PFA_IG_ROWS(kADense, 1, 2);
        } else {
            PFA_IG_ROWS(kADense, 1, 1);
        }
    } else if (N % 64 == 0) {
#if PFA_IG_TILE64 == 1     // experiment (tools/igemm_bench.py): 64 x 64 tiles, 20 KB of LDS per workgroup -> 8 instead of 5 waves per SIMD
        PFA_IG_ROWS_MODE(1, 4)
#elif PFA_IG_TILE64 == 2   // ... or 128 x 32: 25.6 KB, 6 waves per SIMD
        PFA_IG_ROWS_MODE(2, 2)
#else
        PFA_IG_ROWS_MODE(2, 4)
#endif
    } else if (N % 32 == 0) {
        PFA_IG_ROWS_MODE(2, 2)   // (128 x 32 beats 256 x 32: the smaller stage keeps more workgroups per CU)
    } else {
        PFA_IG_ROWS_MODE(4, 1)
    }
#undef PFA_IG_ROWS_MODE
#undef PFA_IG_ROWS
    PFA_LAUNCH_CHECK();
    return 0;
}

extern "C" int pfa_igemm_set_products(int32_t mode) {
    PFA_REQUIRE(mode == 0 || mode == 1, "igemm.set_products: 0 = fp32 MFMA (default), 1 = six bf16 partial products per fp32 product");
    g_ig_products = mode;
    return 0;
}
extern "C" int pfa_igemm_get_products(void) { return g_ig_products; }

extern "C" size_t pfa_igemm_weights_workspace_bytes(int64_t M, int32_t K, int32_t N) {
    if (M < 1 || K < 4 || N < 16 || N % 16 != 0) return 0;
    const IgWeightPlan p = ig_weight_plan(M, K, N);
    return ig_partial_bytes(p, K, N) + (size_t)p.splits * (size_t)N * sizeof(double) + 256;
}

extern "C" int pfa_igemm_weights(const pfa_igemm_operand *a, int64_t M, int32_t K, const float *D, int32_t ldd, int32_t N, float *out,
                                 int32_t perm, int32_t accumulate, float *bias_out, void *workspace, pfa_stream_t stream) {
    if (int rc = ig_check_a(a, M, K)) return rc;
    PFA_REQUIRE(a->mode != kACol2im, "igemm.weights: the col2im operand belongs to the rows form");
    PFA_REQUIRE(M >= 1 && M < (1ll << 31) && D && out && workspace && N >= 16 && N % 16 == 0 && ldd >= N && ldd % 4 == 0 && ((uintptr_t)D & 15) == 0,
                "igemm.weights: bad shapes");
    PFA_REQUIRE(perm >= 0 && perm <= 4, "igemm.weights: bad permutation");
    const IgA A = ig_make_a(a);
    const IgWeightPlan p = ig_weight_plan(M, K, N);
    float *partial = (float *)workspace;
    double *colpart = (double *)((char *)workspace + ig_partial_bytes(p, K, N));
    const dim3 grid(p.tiles, p.splits);
    {
        ScopedKernelTimer timer("igemm_weights", (hipStream_t)stream);
#define PFA_IG_WEIGHTS(MODE, KJ, NI)                                                                                                             \
    hipLaunchKernelGGL((igemm_weights_kernel<MODE, KJ, NI>), grid, dim3(kIgThreads), 0, (hipStream_t)stream, A, (int)M, (int)K, D, (int)ldd, (int)N, \
                       p.rows_per_split, partial, colpart)
#define PFA_IG_WEIGHTS_KJ(MODE, NI)                                  \
    if (p.kj == 4) PFA_IG_WEIGHTS(MODE, 4, NI);                      \
    else if (p.kj == 3) PFA_IG_WEIGHTS(MODE, 3, NI);                 \
    else if (p.kj == 1 && MODE == kADense) PFA_IG_WEIGHTS(kADense, 1, NI); \
    else PFA_IG_WEIGHTS(MODE, 2, NI);
#define PFA_IG_WEIGHTS_MODE(NI)                                        \
    switch (A.mode) {                                                  \
        case kADense: PFA_IG_WEIGHTS_KJ(kADense, NI) break;            \
        case kAIm2colF32: PFA_IG_WEIGHTS_KJ(kAIm2colF32, NI) break;    \
        default: PFA_IG_WEIGHTS_KJ(kAIm2colU8, NI) break;              \
    }
        if (p.tn == 64) {
            PFA_IG_WEIGHTS_MODE(4)
        } else if (p.tn == 32) {
            PFA_IG_WEIGHTS_MODE(2)
        } else {
            PFA_IG_WEIGHTS_MODE(1)
        }
#undef PFA_IG_WEIGHTS_KJ
#undef PFA_IG_WEIGHTS_MODE
#undef PFA_IG_WEIGHTS
        PFA_LAUNCH_CHECK();
    }
    const long long total = (long long)K * N + N;
    hipLaunchKernelGGL(igemm_weights_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, (hipStream_t)stream, partial, colpart, p.splits,
                       (int)K, (int)N, out, bias_out, (int)perm, A.g, (int)accumulate);
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
    PFA_REQUIRE(w && geom && (fwd || dx), "cnn.pack_conv: null buffer");
    const IgA A = ig_make_a(geom);
    PFA_REQUIRE(!dx || (A.g.S >= 1 && A.g.KH % A.g.S == 0 && A.g.KW % A.g.S == 0), "cnn.pack_conv: the dX form needs KH, KW multiples of the stride");
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
