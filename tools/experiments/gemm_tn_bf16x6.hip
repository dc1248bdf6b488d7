// gemm_tn_bf16x6.hip — EXPERIMENT (not product code; DESIGN §8 "what comes next", items 3 and 6): a weight-gradient contraction
//   G[k][n] = sum_m A[m][k] D[m][n]            (fp32 in, fp32 partial sums out; m = minibatch rows, the SLOW index of both operands)
// on the bf16 matrix path.  The bf16 MFMA contracts over 32 rows per instruction and a lane supplies EIGHT consecutive rows of one
// column, so the operands have to reach LDS transposed ([column][row], rows contiguous).  The loader does it on the way in: a thread
// loads the same four columns of two consecutive rows, splits both into three bf16 pieces (hi + mid + lo) and packs the (row, row+1)
// pair of every column and piece into one 32-bit LDS write.  Shape: the recurrent policy's dW_ih (csrc/gemm.hip, 512 x 128 output,
// 131 072 rows, 128 row splits), whose fp32 kernel gemm_tn_partial<2,2,4,4> takes ~152 us per launch (113 TFLOP/s).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/gemm_tn_bf16x6.hip -o tools/experiments/_bin/gemm_tn_bf16x6
// RESULT (MI355X): packing loader 234 us (slower than fp32); row-major planes + ds_read_b64_tr_b16 fragments (second kernel below)
// 131 us = 131 TFLOP/s; both 4.5e-7 of the largest sum from an f64 evaluation.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TK = 128, TN = 128, BM = 32, THREADS = 256;
constexpr int RS = 80;                      // bytes per LDS row: 32 rows (m) of bf16 + 16 bytes of padding
constexpr int PLANE = TK * RS;              // one piece of one operand: [column][m]

__device__ __forceinline__ uint32_t pk(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<uint32_t *>(&r);
}
__device__ __forceinline__ float lo_f(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_f(uint32_t p) { return __uint_as_float(p & 0xFFFF0000u); }

// pieces of one value pair (two ROWS of the same column): out[p] = bf16 piece p of `a` (low half) | piece p of `b` (high half)
__device__ __forceinline__ void split_pair(float a, float b, uint32_t (&out)[3]) {
    out[0] = pk(a, b);
    const float ra = a - lo_f(out[0]), rb = b - hi_f(out[0]);
    out[1] = pk(ra, rb);
    out[2] = pk(ra - lo_f(out[1]), rb - hi_f(out[1]));
}

__global__ void __launch_bounds__(THREADS) gemm_tn_bf16x6_kernel(const float *__restrict__ A, int lda, const float *__restrict__ D, int ldd, int K, int N,
                                                                 long long rows_per_split, float *__restrict__ partial) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 3 * PLANE];     // operand (A, D) x piece x [128 columns][32 rows]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, g = lane >> 4;
    const int wr = wv >> 1, wc = wv & 1;                                          // wave (wr, wc) owns the 64 x 64 quarter of the tile
    const int tiles_n = N / TN;
    const int k0 = (blockIdx.x / tiles_n) * TK, n0 = (blockIdx.x % tiles_n) * TN;
    const long long m_lo = (long long)blockIdx.y * rows_per_split;
    const int slabs = (int)(rows_per_split / BM);
    // loader: columns 4 col4 .. +3 of the row pairs (2 rp, 2 rp + 1) and (2 rp + 16, 2 rp + 17)
    const int col4 = tid & 31, rp = tid >> 5;
    const float *ap = A + (m_lo + 2 * rp) * lda + k0 + 4 * col4, *dp = D + (m_lo + 2 * rp) * ldd + n0 + 4 * col4;
    float4 a0, a1, a2, a3, d0, d1, d2, d3;
    auto fetch = [&]() {
        a0 = *reinterpret_cast<const float4 *>(ap);
        a1 = *reinterpret_cast<const float4 *>(ap + lda);
        a2 = *reinterpret_cast<const float4 *>(ap + 16 * (size_t)lda);
        a3 = *reinterpret_cast<const float4 *>(ap + 17 * (size_t)lda);
        d0 = *reinterpret_cast<const float4 *>(dp);
        d1 = *reinterpret_cast<const float4 *>(dp + ldd);
        d2 = *reinterpret_cast<const float4 *>(dp + 16 * (size_t)ldd);
        d3 = *reinterpret_cast<const float4 *>(dp + 17 * (size_t)ldd);
        ap += (size_t)BM * lda;
        dp += (size_t)BM * ldd;
    };
    auto put = [&](unsigned char *base, const float4 &x, const float4 &y, int row) {   // rows (row, row + 1) of four columns
        uint32_t p[3];
        unsigned char *q = base + (4 * col4) * RS + row * 2;
        split_pair(x.x, y.x, p);
        *reinterpret_cast<uint32_t *>(q) = p[0], *reinterpret_cast<uint32_t *>(q + PLANE) = p[1], *reinterpret_cast<uint32_t *>(q + 2 * PLANE) = p[2];
        split_pair(x.y, y.y, p);
        *reinterpret_cast<uint32_t *>(q + RS) = p[0], *reinterpret_cast<uint32_t *>(q + RS + PLANE) = p[1], *reinterpret_cast<uint32_t *>(q + RS + 2 * PLANE) = p[2];
        split_pair(x.z, y.z, p);
        *reinterpret_cast<uint32_t *>(q + 2 * RS) = p[0], *reinterpret_cast<uint32_t *>(q + 2 * RS + PLANE) = p[1], *reinterpret_cast<uint32_t *>(q + 2 * RS + 2 * PLANE) = p[2];
        split_pair(x.w, y.w, p);
        *reinterpret_cast<uint32_t *>(q + 3 * RS) = p[0], *reinterpret_cast<uint32_t *>(q + 3 * RS + PLANE) = p[1], *reinterpret_cast<uint32_t *>(q + 3 * RS + 2 * PLANE) = p[2];
    };
    auto stash = [&]() {
        put(lds, a0, a1, 2 * rp);
        put(lds, a2, a3, 2 * rp + 16);
        put(lds + 3 * PLANE, d0, d1, 2 * rp);
        put(lds + 3 * PLANE, d2, d3, 2 * rp + 16);
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char *la = lds + (wr * 64 + c) * RS + g * 16, *lb = lds + 3 * PLANE + (wc * 64 + c) * RS + g * 16;
    fetch();
    for (int s = 0; s < slabs; ++s) {
        stash();
        __syncthreads();
        if (s + 1 < slabs) fetch();
        // (piece of A, piece of D): 0 = hi, 1 = mid, 2 = lo; small terms first; fragments of one piece pair at a time (registers)
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const bf16x8 *>(la + TA[t] * PLANE + a * 16 * RS);
#pragma unroll
            for (int b = 0; b < 4; ++b) fb[b] = *reinterpret_cast<const bf16x8 *>(lb + TB[t] * PLANE + b * 16 * RS);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }
    float *out = partial + (size_t)blockIdx.y * K * N;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(k0 + wr * 64 + a * 16 + 4 * g + r) * N + n0 + wc * 64 + b * 16 + c] = acc[a][b][r];
}

// ---- variant: row-major bf16 planes in LDS ([32 rows][128 columns], the layout a plain loader writes with 8-byte stores) and the
// transposition done by the LDS itself: ds_read_b64_tr_b16 (lane map: tools/experiments/ds_read_tr_map.hip) hands lane n of a 16-lane
// group four consecutive ROWS of column n; two reads make one 16 x 16 x 32 fragment.
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int RSB = TK * 2 + 8;             // bytes per LDS row: 128 bf16 + 8 bytes of padding
constexpr int PLANEB = BM * RSB;

__device__ __forceinline__ void split4(const float4 &x, uint2 &h, uint2 &m, uint2 &l) {
    h.x = pk(x.x, x.y);
    h.y = pk(x.z, x.w);
    const float r0 = x.x - lo_f(h.x), r1 = x.y - hi_f(h.x), r2 = x.z - lo_f(h.y), r3 = x.w - hi_f(h.y);
    m.x = pk(r0, r1);
    m.y = pk(r2, r3);
    l.x = pk(r0 - lo_f(m.x), r1 - hi_f(m.x));
    l.y = pk(r2 - lo_f(m.y), r3 - hi_f(m.y));
}
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char *p) {     // rows 8 g .. 8 g + 7 of this lane's column
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    union {
        s16x4 h[2];
        bf16x8 v;
    } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)p);
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p + 4 * RSB));
    return u.v;
}

__global__ void __launch_bounds__(THREADS) gemm_tn_bf16x6_tr_kernel(const float *__restrict__ A, int lda, const float *__restrict__ D, int ldd, int K,
                                                                    int N, long long rows_per_split, float *__restrict__ partial) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 3 * PLANEB];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, g = lane >> 4;
    const int wr = wv >> 1, wc = wv & 1;
    const int tiles_n = N / TN;
    const int k0 = (blockIdx.x / tiles_n) * TK, n0 = (blockIdx.x % tiles_n) * TN;
    const long long m_lo = (long long)blockIdx.y * rows_per_split;
    const int slabs = (int)(rows_per_split / BM);
    const int col4 = tid & 31, rr = tid >> 5;                 // loader: rows rr + 8 q, columns 4 col4 .. + 3
    const float *ap = A + (m_lo + rr) * lda + k0 + 4 * col4, *dp = D + (m_lo + rr) * ldd + n0 + 4 * col4;
    float4 a0, a1, a2, a3, d0, d1, d2, d3;
    auto fetch = [&]() {
        a0 = *reinterpret_cast<const float4 *>(ap);
        a1 = *reinterpret_cast<const float4 *>(ap + 8 * (size_t)lda);
        a2 = *reinterpret_cast<const float4 *>(ap + 16 * (size_t)lda);
        a3 = *reinterpret_cast<const float4 *>(ap + 24 * (size_t)lda);
        d0 = *reinterpret_cast<const float4 *>(dp);
        d1 = *reinterpret_cast<const float4 *>(dp + 8 * (size_t)ldd);
        d2 = *reinterpret_cast<const float4 *>(dp + 16 * (size_t)ldd);
        d3 = *reinterpret_cast<const float4 *>(dp + 24 * (size_t)ldd);
        ap += (size_t)BM * lda;
        dp += (size_t)BM * ldd;
    };
    auto put = [&](unsigned char *base, const float4 &x, int row) {
        uint2 h, m, l;
        split4(x, h, m, l);
        unsigned char *q = base + row * RSB + col4 * 8;
        *reinterpret_cast<uint2 *>(q) = h;
        *reinterpret_cast<uint2 *>(q + PLANEB) = m;
        *reinterpret_cast<uint2 *>(q + 2 * PLANEB) = l;
    };
    auto stash = [&]() {
        put(lds, a0, rr), put(lds, a1, rr + 8), put(lds, a2, rr + 16), put(lds, a3, rr + 24);
        put(lds + 3 * PLANEB, d0, rr), put(lds + 3 * PLANEB, d1, rr + 8), put(lds + 3 * PLANEB, d2, rr + 16), put(lds + 3 * PLANEB, d3, rr + 24);
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this lane's address inside a 16-column fragment: row 8 g + (c >> 2), columns 4 (c & 3) .. + 3 (c = lane within the 16-lane group)
    const unsigned char *la = lds + (8 * g + (c >> 2)) * RSB + (wr * 64 + 4 * (c & 3)) * 2;
    const unsigned char *lb = lds + 3 * PLANEB + (8 * g + (c >> 2)) * RSB + (wc * 64 + 4 * (c & 3)) * 2;
    fetch();
    for (int s = 0; s < slabs; ++s) {
        stash();
        __syncthreads();
        if (s + 1 < slabs) fetch();
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) fa[a] = frag_tr(la + TA[t] * PLANEB + a * 32);
#pragma unroll
            for (int b = 0; b < 4; ++b) fb[b] = frag_tr(lb + TB[t] * PLANEB + b * 32);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }
    float *out = partial + (size_t)blockIdx.y * K * N;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(k0 + wr * 64 + a * 16 + 4 * g + r) * N + n0 + wc * 64 + b * 16 + c] = acc[a][b][r];
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main() {
    const int K = 512, N = 128, splits = 128;
    const long long M = 131072, per = M / splits;          // 1024 rows per split
    std::vector<float> a((size_t)M * K), d((size_t)M * N);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() {
        st ^= st << 13, st ^= st >> 7, st ^= st << 17;
        return (float)((st >> 11) * (1.0 / 9007199254740992.0));
    };
    for (auto &v : a) v = (rnd() - 0.5f) * 0.02f;           // gradient-like magnitudes
    for (auto &v : d) {
        const float u = rnd();
        v = u < 0.5f ? 0.0f : 2.0f * (u - 0.5f);            // post-ReLU-like activations
    }
    float *dA, *dD, *dP;
    CK(hipMalloc(&dA, a.size() * 4));
    CK(hipMalloc(&dD, d.size() * 4));
    CK(hipMalloc(&dP, (size_t)splits * K * N * 4));
    CK(hipMemcpy(dA, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dD, d.data(), d.size() * 4, hipMemcpyHostToDevice));
    const dim3 grid((K / TK) * (N / TN), splits);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_tn_bf16x6_kernel, grid, dim3(THREADS), 0, 0, dA, K, dD, N, K, N, per, dP);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_tn_bf16x6_kernel, grid, dim3(THREADS), 0, 0, dA, K, dD, N, K, N, per, dP);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, flop = 2.0 * M * N * K;
    std::vector<float> p((size_t)splits * K * N);
    CK(hipMemcpy(p.data(), dP, p.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0, scale = 0.0;
    for (int s = 0; s < 96; ++s) {
        const int k = (int)((s * 2654435761u) % K), n = (int)((s * 40503u + 17) % N);
        double ref = 0.0, got = 0.0;
        for (long long m = 0; m < M; ++m) ref += (double)a[(size_t)m * K + k] * (double)d[(size_t)m * N + n];
        for (int q = 0; q < splits; ++q) got += (double)p[((size_t)q * K + k) * N + n];
        worst = fmax(worst, fabs(got - ref));
        scale = fmax(scale, fabs(ref));
    }
    // ---- the transpose-read variant
    float *dP2;
    CK(hipMalloc(&dP2, (size_t)splits * K * N * 4));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_tn_bf16x6_tr_kernel, grid, dim3(THREADS), 0, 0, dA, K, dD, N, K, N, per, dP2);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_tn_bf16x6_tr_kernel, grid, dim3(THREADS), 0, 0, dA, K, dD, N, K, N, per, dP2);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us2 = ms * 1e3 / reps;
    std::vector<float> p2((size_t)splits * K * N);
    CK(hipMemcpy(p2.data(), dP2, p2.size() * 4, hipMemcpyDeviceToHost));
    double worst2 = 0.0;
    for (int s = 0; s < 96; ++s) {
        const int k = (int)((s * 2654435761u) % K), n = (int)((s * 40503u + 17) % N);
        double ref = 0.0, got = 0.0;
        for (long long m = 0; m < M; ++m) ref += (double)a[(size_t)m * K + k] * (double)d[(size_t)m * N + n];
        for (int q = 0; q < splits; ++q) got += (double)p2[((size_t)q * K + k) * N + n];
        worst2 = fmax(worst2, fabs(got - ref));
    }
    printf("row-major planes + ds_read_b64_tr_b16 fragments: %.1f us per launch = %.1f TFLOP/s; max |err| / max |exact|: %.2e\n", us2, flop / us2 / 1e6,
           worst2 / scale);
    printf("gemm_tn_bf16x6  G[%d][%d] over %lld rows, %d splits: %.1f us per launch = %.1f TFLOP/s of the fp32 product it replaces (the product's fp32\n"
           "                kernel gemm_tn_partial<2,2,4,4>: ~152 us = 113 TFLOP/s); max |err| of 96 sampled sums / max |exact|: %.2e\n",
           K, N, M, splits, us, flop / us / 1e6, worst / scale);
    return 0;
}
