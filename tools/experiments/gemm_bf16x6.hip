// gemm_bf16x6.hip — EXPERIMENT (not product code; DESIGN §8 "what comes next", item 3): an fp32 GEMM on the bf16 matrix path.
//   C[m][n] = sum_k A[m][k] B[n][k]        fp32 in, fp32 out, the shape of the NatureCNN's Linear(3136, 512) over an 8192-frame chunk
// Every fp32 operand is split into three bf16 pieces in the loader (a = hi + mid + lo, round-to-nearest each time), the pieces go
// through LDS as three bf16 planes, and every 16 x 16 x 32 tile product is issued as the six partial products above 2^-24 relative
// (lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi — small terms first) on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.
// tools/experiments/bf16_split_accuracy.py is the host-side accuracy study; this file measures what it costs on the device:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/gemm_bf16x6.hip -o tools/experiments/_bin/gemm_bf16x6
//   tools/experiments/_bin/gemm_bf16x6            (on the GPU box: prints the time per launch, TFLOP/s of the fp32 product it replaces,
//                                                  and the largest error of sampled outputs against an f64 dot product)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 128, TN = 64, BK = 32, THREADS = 256;
constexpr int ROWS = TM + TN;
constexpr int RS = 80;                       // bytes per LDS row of a plane: 32 bf16 + 16 bytes of padding
constexpr int PLANE = ROWS * RS;             // (the 128 x 64 tile; the in-loader kernel is also instantiated at 64 x 64)

__device__ __forceinline__ uint32_t pk(float a, float b) {   // two floats -> two bf16 (round to nearest even), a in the low half
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<uint32_t *>(&r);
}
__device__ __forceinline__ float lo_f(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_f(uint32_t p) { return __uint_as_float(p & 0xFFFF0000u); }

// x = hi + mid + lo, four values at a time: three 8-byte groups of four bf16
__device__ __forceinline__ void split4(const float4 &x, uint2 &h, uint2 &m, uint2 &l) {
    h.x = pk(x.x, x.y);
    h.y = pk(x.z, x.w);
    const float r0 = x.x - lo_f(h.x), r1 = x.y - hi_f(h.x), r2 = x.z - lo_f(h.y), r3 = x.w - hi_f(h.y);
    m.x = pk(r0, r1);
    m.y = pk(r2, r3);
    l.x = pk(r0 - lo_f(m.x), r1 - hi_f(m.x));
    l.y = pk(r2 - lo_f(m.y), r3 - hi_f(m.y));
}

template <int MI>
__global__ void __launch_bounds__(THREADS) gemm_bf16x6_kernel(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M,
                                                              int N, int K) {
    constexpr int TM = 64 * MI, ROWS = TM + TN, PLANE = ROWS * RS;
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
    const int lr = tid >> 3, kq = (tid & 7) * 4;                 // loader: rows lr + 32 q, four consecutive k
    const float *ap = A + (size_t)(m0 + lr) * K + kq, *bp = B + (size_t)(n0 + lr) * K + kq;
    float4 ra[2 * MI], rb[2];
    auto fetch = [&]() {
#pragma unroll
        for (int q = 0; q < 2 * MI; ++q) ra[q] = *reinterpret_cast<const float4 *>(ap + (size_t)(32 * q) * K);
#pragma unroll
        for (int q = 0; q < 2; ++q) rb[q] = *reinterpret_cast<const float4 *>(bp + (size_t)(32 * q) * K);
        ap += BK;
        bp += BK;
    };
    auto stash = [&]() {
#pragma unroll
        for (int q = 0; q < 2 * MI; ++q) {
            uint2 h, m, l;
            split4(ra[q], h, m, l);
            unsigned char *p = lds + (lr + 32 * q) * RS + kq * 2;
            *reinterpret_cast<uint2 *>(p) = h;
            *reinterpret_cast<uint2 *>(p + PLANE) = m;
            *reinterpret_cast<uint2 *>(p + 2 * PLANE) = l;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            uint2 h, m, l;
            split4(rb[q], h, m, l);
            unsigned char *p = lds + (TM + lr + 32 * q) * RS + kq * 2;
            *reinterpret_cast<uint2 *>(p) = h;
            *reinterpret_cast<uint2 *>(p + PLANE) = m;
            *reinterpret_cast<uint2 *>(p + 2 * PLANE) = l;
        }
    };
    f32x4 acc[MI][4];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char *la = lds + (wv * 16 * MI + c) * RS + g * 16, *lb = lds + (TM + c) * RS + g * 16;
    const int slabs = K / BK;
    fetch();
    for (int s = 0; s < slabs; ++s) {
        stash();
        __syncthreads();
        if (s + 1 < slabs) fetch();          // the next slab's global loads fly under this slab's products
        bf16x8 fa[3][MI], fb[3][4];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int a = 0; a < MI; ++a) fa[p][a] = *reinterpret_cast<const bf16x8 *>(la + p * PLANE + a * 16 * RS);
#pragma unroll
            for (int b = 0; b < 4; ++b) fb[p][b] = *reinterpret_cast<const bf16x8 *>(lb + p * PLANE + b * 16 * RS);
        }
        // (piece of A, piece of B): 0 = hi, 1 = mid, 2 = lo; small terms first
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[TA[t]][a], fb[TB[t]][b], acc[a][b], 0, 0, 0);
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wv * 16 * MI + a * 16 + 4 * g + r;
#pragma unroll
            for (int b = 0; b < 4; ++b) C[(size_t)m * N + n0 + b * 16 + c] = acc[a][b][r];
        }
}

// ---- variant: the operands arrive already split (what a producing layer's epilogue / the weight pack would write): three bf16 planes
// [3][rows][K]; the loader moves 16-byte groups of eight bf16 from global memory to LDS with no arithmetic at all.
__global__ void __launch_bounds__(256) split_planes_kernel(const float *__restrict__ x, long long n4, long long plane_elems, __bf16 *__restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    uint2 h, m, l;
    split4(reinterpret_cast<const float4 *>(x)[i], h, m, l);
    *reinterpret_cast<uint2 *>(out + 4 * i) = h;
    *reinterpret_cast<uint2 *>(out + plane_elems + 4 * i) = m;
    *reinterpret_cast<uint2 *>(out + 2 * plane_elems + 4 * i) = l;
}

__global__ void __launch_bounds__(THREADS) gemm_bf16x6_pre_kernel(const __bf16 *__restrict__ A3, const __bf16 *__restrict__ B3, float *__restrict__ C,
                                                                  int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][3 * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
    const int lr = tid >> 2, k8 = (tid & 3) * 8;                 // loader: rows lr + 64 q, eight consecutive k of each plane
    const size_t pa = (size_t)M * K, pb = (size_t)N * K;
    const __bf16 *ap = A3 + (size_t)(m0 + lr) * K + k8, *bp = B3 + (size_t)(n0 + lr) * K + k8;
    // (named registers, not arrays: an array written inside these lambdas is promoted to LDS by the compiler — 36 KB of it)
    uint4 a00, a01, a10, a11, a20, a21, b0, b1, b2;
    auto fetch = [&]() {
        a00 = *reinterpret_cast<const uint4 *>(ap);
        a01 = *reinterpret_cast<const uint4 *>(ap + (size_t)64 * K);
        a10 = *reinterpret_cast<const uint4 *>(ap + pa);
        a11 = *reinterpret_cast<const uint4 *>(ap + pa + (size_t)64 * K);
        a20 = *reinterpret_cast<const uint4 *>(ap + 2 * pa);
        a21 = *reinterpret_cast<const uint4 *>(ap + 2 * pa + (size_t)64 * K);
        b0 = *reinterpret_cast<const uint4 *>(bp);
        b1 = *reinterpret_cast<const uint4 *>(bp + pb);
        b2 = *reinterpret_cast<const uint4 *>(bp + 2 * pb);
        ap += BK;
        bp += BK;
    };
    auto stash = [&](int buf) {
        unsigned char *pa_ = lds[buf] + lr * RS + k8 * 2, *pb_ = lds[buf] + (TM + lr) * RS + k8 * 2;
        *reinterpret_cast<uint4 *>(pa_) = a00;
        *reinterpret_cast<uint4 *>(pa_ + 64 * RS) = a01;
        *reinterpret_cast<uint4 *>(pa_ + PLANE) = a10;
        *reinterpret_cast<uint4 *>(pa_ + PLANE + 64 * RS) = a11;
        *reinterpret_cast<uint4 *>(pa_ + 2 * PLANE) = a20;
        *reinterpret_cast<uint4 *>(pa_ + 2 * PLANE + 64 * RS) = a21;
        *reinterpret_cast<uint4 *>(pb_) = b0;
        *reinterpret_cast<uint4 *>(pb_ + PLANE) = b1;
        *reinterpret_cast<uint4 *>(pb_ + 2 * PLANE) = b2;
    };
    f32x4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int slabs = K / BK;
    fetch();
    stash(0);
    __syncthreads();
    for (int s = 0; s < slabs; ++s) {
        const int buf = s & 1;
        if (s + 1 < slabs) fetch();
        const unsigned char *la = lds[buf] + (wv * 32 + c) * RS + g * 16, *lb = lds[buf] + (TM + c) * RS + g * 16;
        bf16x8 fa[3][2], fb[3][4];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[p][a] = *reinterpret_cast<const bf16x8 *>(la + p * PLANE + a * 16 * RS);
#pragma unroll
            for (int b = 0; b < 4; ++b) fb[p][b] = *reinterpret_cast<const bf16x8 *>(lb + p * PLANE + b * 16 * RS);
        }
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[TA[t]][a], fb[TB[t]][b], acc[a][b], 0, 0, 0);
        if (s + 1 < slabs) stash(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wv * 32 + a * 16 + 4 * g + r;
#pragma unroll
            for (int b = 0; b < 4; ++b) C[(size_t)m * N + n0 + b * 16 + c] = acc[a][b][r];
        }
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
    const int M = 8192, N = 512, K = 3136;
    std::vector<float> a((size_t)M * K), b((size_t)N * K);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() {
        st ^= st << 13, st ^= st >> 7, st ^= st << 17;
        return (float)((st >> 11) * (1.0 / 9007199254740992.0));
    };
    for (auto &v : a) {                       // post-ReLU-like activations: half zeros, the rest in (0, 2)
        const float u = rnd();
        v = u < 0.5f ? 0.0f : 4.0f * (u - 0.5f);
    }
    for (auto &v : b) v = (rnd() - 0.5f) * 2.0f * sqrtf(6.0f / K);
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, a.size() * 4));
    CK(hipMalloc(&dB, b.size() * 4));
    CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    const dim3 grid(M / TM, N / TN), grid1(M / 64, N / TN);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_bf16x6_kernel<2>, grid, dim3(THREADS), 0, 0, dA, dB, dC, M, N, K);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_bf16x6_kernel<2>, grid, dim3(THREADS), 0, 0, dA, dB, dC, M, N, K);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    float ms1 = 0.f;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_bf16x6_kernel<1>, grid1, dim3(THREADS), 0, 0, dA, dB, dC, M, N, K);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_bf16x6_kernel<1>, grid1, dim3(THREADS), 0, 0, dA, dB, dC, M, N, K);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms1, e0, e1));
    printf("64 x 64 tiles (1024 workgroups): %.1f us per launch\n", ms1 * 1e3 / reps);
    const double us = ms * 1e3 / reps, flop = 2.0 * M * N * K;
    std::vector<float> cout_((size_t)M * N);
    CK(hipMemcpy(cout_.data(), dC, cout_.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0, scale = 0.0, worst32 = 0.0;
    for (int s = 0; s < 256; ++s) {
        const int m = (int)((s * 2654435761u) % M), n = (int)((s * 40503u + 17) % N);
        double ref = 0.0;
        float chain = 0.0f;
        for (int k = 0; k < K; ++k) {
            ref += (double)a[(size_t)m * K + k] * (double)b[(size_t)n * K + k];
            chain = fmaf(a[(size_t)m * K + k], b[(size_t)n * K + k], chain);
        }
        worst = fmax(worst, fabs((double)cout_[(size_t)m * N + n] - ref));
        worst32 = fmax(worst32, fabs((double)chain - ref));
        scale = fmax(scale, fabs(ref));
    }
    // ---- the pre-split variant
    __bf16 *dA3, *dB3;
    float *dC2;
    CK(hipMalloc(&dA3, a.size() * 2 * 3));
    CK(hipMalloc(&dB3, b.size() * 2 * 3));
    CK(hipMalloc(&dC2, (size_t)M * N * 4));
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((a.size() / 4 + 255) / 256)), dim3(256), 0, 0, dA, (long long)(a.size() / 4), (long long)a.size(), dA3);
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((b.size() / 4 + 255) / 256)), dim3(256), 0, 0, dB, (long long)(b.size() / 4), (long long)b.size(), dB3);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_bf16x6_pre_kernel, grid, dim3(THREADS), 0, 0, dA3, dB3, dC2, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_bf16x6_pre_kernel, grid, dim3(THREADS), 0, 0, dA3, dB3, dC2, M, N, K);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us2 = ms * 1e3 / reps;
    std::vector<float> c2((size_t)M * N);
    CK(hipMemcpy(c2.data(), dC2, c2.size() * 4, hipMemcpyDeviceToHost));
    double diff = 0.0;
    for (size_t i = 0; i < c2.size(); ++i) diff = fmax(diff, fabs((double)c2[i] - (double)cout_[i]));
    printf("pre-split operands (three bf16 planes in memory, no arithmetic in the loader, double-buffered LDS): %.1f us per launch = %.1f TFLOP/s;\n"
           "             max |difference| to the in-loader split: %.2e\n", us2, flop / us2 / 1e6, diff);
    printf("gemm_bf16x6  M=%d N=%d K=%d: %.1f us per launch = %.1f TFLOP/s of the fp32 product it replaces (fp32-MFMA peak 157.3; the product's\n"
           "             fp32 kernel, igemm_rows<dense,128x64>: ~252-270 us = 97-104 TFLOP/s)\n"
           "             max |err| over 256 sampled outputs / max |exact|: six-term split %.2e, sequential fp32 fma chain %.2e\n",
           M, N, K, us, flop / us / 1e6, worst / scale, worst32 / scale);
    return 0;
}
