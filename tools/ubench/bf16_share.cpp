// Developer microbenchmark (not product code): do bf16 MFMAs (v_mfma_f32_16x16x32_bf16 / 16x16x16) share a SIMD with VALU work, or
// does their time ADD like the fp32 MFMA's does (simd_share.cpp, profiles/r02_ubench_simd_share.txt)?  That decides whether the
// operand-splitting VALU work of a "three bf16 pieces per fp32 operand" product form can hide under its own matrix instructions.
// hipcc --offload-arch=gfx950 -O3 bf16_share.cpp -o bf16_share
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define VFMA(v, a, b) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b))
#define VAND(v, a) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v) : "v"(a))
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// roles: 0 idle   1 fp32 MFMA 16x16x4 (4 chains)   2 VALU fma (8 chains)   3 bf16 MFMA 16x16x32 (4 chains)   4 bf16 MFMA 16x16x16 (4 chains)
//        5 bf16 MFMA 16x16x32 + VALU fma interleaved in ONE stream (4 fma per MFMA)   6 VALU integer ops (v_and, 8 chains)
//        7 fp32 MFMA + integer VALU interleaved in one stream   8 v_cvt_pk_bf16_f32 stream
// Per iteration: 16 MFMAs (roles 1/3/4/5/7), 256 VALU ops (roles 2/6/8), 64 VALU ops beside the 16 MFMAs (roles 5/7).
template <int RA, int RB>
__global__ void __launch_bounds__(512, 2) share_kernel(float *out, int iters, float seed) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = wv < 4 ? RA : RB;
    float acc = 0.f;
    if (role == 1 || role == 7) {
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        float a = seed + lane, b = seed * 2 + lane;
        int v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3, m = 0x7fffffff - lane;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
                if (role == 7) { VAND(v0, m); VAND(v1, m); VAND(v2, m); VAND(v3, m); }
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
                if (role == 7) { VAND(v0, m); VAND(v1, m); VAND(v2, m); VAND(v3, m); }
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
                if (role == 7) { VAND(v0, m); VAND(v1, m); VAND(v2, m); VAND(v3, m); }
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
                if (role == 7) { VAND(v0, m); VAND(v1, m); VAND(v2, m); VAND(v3, m); }
            }
        }
        acc = c0[0] + c1[1] + c2[2] + c3[3] + (float)(v0 + v1 + v2 + v3);
    } else if (role == 3 || role == 5) {
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        bf16x8 a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + lane + i); b[i] = (__bf16)(seed * 2 + lane - i); }
        float fa = seed + lane, fb = seed;
        float v0 = seed, v1 = seed + 1, v2 = seed + 2, v3 = seed + 3, v4 = seed + 4, v5 = seed + 5, v6 = seed + 6, v7 = seed + 7;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                if (role == 5) { VFMA(v0, fa, fb); VFMA(v1, fa, fb); VFMA(v2, fa, fb); VFMA(v3, fa, fb); }
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
                if (role == 5) { VFMA(v4, fa, fb); VFMA(v5, fa, fb); VFMA(v6, fa, fb); VFMA(v7, fa, fb); }
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
                if (role == 5) { VFMA(v0, fa, fb); VFMA(v1, fa, fb); VFMA(v2, fa, fb); VFMA(v3, fa, fb); }
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
                if (role == 5) { VFMA(v4, fa, fb); VFMA(v5, fa, fb); VFMA(v6, fa, fb); VFMA(v7, fa, fb); }
            }
        }
        acc = c0[0] + c1[1] + c2[2] + c3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    } else if (role == 4) {
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        s16x4 a, b;
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = (short)(0x3f80 + lane + i); b[i] = (short)(0x3f80 + lane - i); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c3, 0, 0, 0);
            }
        }
        acc = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (role == 2) {
        float v0 = seed, v1 = seed + 1, v2 = seed + 2, v3 = seed + 3, v4 = seed + 4, v5 = seed + 5, v6 = seed + 6, v7 = seed + 7;
        const float a = 1.0f + seed, b = seed;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                VFMA(v0, a, b); VFMA(v1, a, b); VFMA(v2, a, b); VFMA(v3, a, b);
                VFMA(v4, a, b); VFMA(v5, a, b); VFMA(v6, a, b); VFMA(v7, a, b);
            }
        }
        acc = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    } else if (role == 6) {
        int v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3, v4 = lane + 4, v5 = lane + 5, v6 = lane + 6, v7 = lane + 7, m = 0x7fffffff - lane;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                VAND(v0, m); VAND(v1, m); VAND(v2, m); VAND(v3, m);
                VAND(v4, m); VAND(v5, m); VAND(v6, m); VAND(v7, m);
            }
        }
        acc = (float)(v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7);
    } else if (role == 8) {
        float x0 = seed + lane, x1 = seed - lane;
        unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r0) : "v"(x0), "v"(x1));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r1) : "v"(x0), "v"(x1));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r2) : "v"(x0), "v"(x1));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r3) : "v"(x0), "v"(x1));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r4) : "v"(x0), "v"(x1));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r5) : "v"(x0), "v"(x1));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r6) : "v"(x0), "v"(x1));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r7) : "v"(x0), "v"(x1));
            }
        }
        acc = (float)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7);
    }
    if (acc == 1234.5678f) out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int RA, int RB>
static float run_share(float *out, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((share_kernel<RA, RB>), dim3(256), dim3(512), 0, 0, out, iters, 0.001f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((share_kernel<RA, RB>), dim3(256), dim3(512), 0, 0, out, iters, 0.001f);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5 * 1e3f;
}

int main() {
    float *out; CK(hipMalloc(&out, 256 * 512 * 4));
    const int it = 2000;   // 2000 x 16 MFMAs per wave: fp32 16x16x4 at 32 cycles each = 1.02 M cycles ~ 0.43 ms at 2.4 GHz
    printf("fp32 MFMA alone                    %8.1f us\n", run_share<1, 0>(out, it));
    printf("VALU fma alone                     %8.1f us\n", run_share<0, 2>(out, it));
    printf("VALU int alone                     %8.1f us\n", run_share<0, 6>(out, it));
    printf("cvt_pk_bf16 alone                  %8.1f us\n", run_share<0, 8>(out, it));
    printf("fp32 MFMA | VALU fma               %8.1f us\n", run_share<1, 2>(out, it));
    printf("fp32 MFMA | VALU int               %8.1f us\n", run_share<1, 6>(out, it));
    printf("fp32 MFMA + VALU int, one stream   %8.1f us\n", run_share<7, 0>(out, it));
    printf("bf16 16x16x32 alone                %8.1f us\n", run_share<3, 0>(out, it));
    printf("bf16 16x16x32 | bf16 16x16x32      %8.1f us\n", run_share<3, 3>(out, it));
    printf("bf16 16x16x16 alone                %8.1f us\n", run_share<4, 0>(out, it));
    printf("bf16 16x16x32 | VALU fma           %8.1f us\n", run_share<3, 2>(out, it));
    printf("bf16 16x16x32 | VALU int           %8.1f us\n", run_share<3, 6>(out, it));
    printf("bf16 16x16x32 | cvt_pk_bf16        %8.1f us\n", run_share<3, 8>(out, it));
    printf("bf16 16x16x32 | fp32 MFMA          %8.1f us\n", run_share<3, 1>(out, it));
    printf("bf16 16x16x32 + VALU fma, 1 stream %8.1f us\n", run_share<5, 0>(out, it));
    printf("(bf16 + VALU fma) x 2 waves        %8.1f us\n", run_share<5, 5>(out, it));
    return 0;
}
