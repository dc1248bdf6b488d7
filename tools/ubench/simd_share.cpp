// Developer microbenchmarks (not product code): what two waves on one SIMD share for f32 MFMA vs VALU vs LDS latency,
// and what a split-K gradient reduction through atomics costs.  hipcc --offload-arch=gfx950 -O3 simd_share.cpp -o simd_share
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define VFMA(v, a, b) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b))
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// roles: 0 idle, 1 MFMA (4 independent chains), 2 VALU fma (8 chains), 3 transcendental exp, 4 LDS bpermute dependent chain,
//        5 MFMA+VALU interleaved in ONE stream, 6 ds_read_b128 stream
template <int RA, int RB, int PRIO = 0>
__global__ void __launch_bounds__(512, 2) share_kernel(float *out, int iters, float seed) {
    __shared__ float lds[8192];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = wv < 4 ? RA : RB;
    if (PRIO == 1 && role != 1) __builtin_amdgcn_s_setprio(3);   // the non-MFMA role above the MFMA role
    if (PRIO == 2 && role == 1) __builtin_amdgcn_s_setprio(3);
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = seed * i;
    __syncthreads();
    float acc = 0.f;
    if (role == 1 || role == 5) {
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        float a = seed + lane, b = seed * 2 + lane;
        float v0 = seed, v1 = seed + 1, v2 = seed + 2, v3 = seed + 3, v4 = seed + 4, v5 = seed + 5, v6 = seed + 6, v7 = seed + 7;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
                if (role == 5) { VFMA(v0, a, b); VFMA(v1, a, b); VFMA(v2, a, b); VFMA(v3, a, b); }
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
                if (role == 5) { VFMA(v4, a, b); VFMA(v5, a, b); VFMA(v6, a, b); VFMA(v7, a, b); }
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
                if (role == 5) { VFMA(v0, a, b); VFMA(v1, a, b); VFMA(v2, a, b); VFMA(v3, a, b); }
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
                if (role == 5) { VFMA(v4, a, b); VFMA(v5, a, b); VFMA(v6, a, b); VFMA(v7, a, b); }
            }
        }
        acc = c0[0] + c1[1] + c2[2] + c3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    } else if (role == 2) {   // 16 MFMA-equivalents of time per iteration would be 16*32 = 512 cycles; issue 256 fma = 512 cycles
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
    } else if (role == 3) {
        float v0 = seed, v1 = seed + 1, v2 = seed + 2, v3 = seed + 3;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                v0 = __builtin_amdgcn_exp2f(v0); v1 = __builtin_amdgcn_exp2f(v1); v2 = __builtin_amdgcn_exp2f(v2); v3 = __builtin_amdgcn_exp2f(v3);
            }
        }
        acc = v0 + v1 + v2 + v3;
    } else if (role == 4) {
        int idx = (lane * 17 + 3) & 63;
        float v = seed + lane;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v = __shfl(v, idx, 64) + 1.0f;   // ds_bpermute dependent chain
        }
        acc = v;
    } else if (role == 6) {
        f32x4 s = {0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) s += *reinterpret_cast<const f32x4 *>(lds + ((lane * 4 + u * 256 + it * 4) & 8188));
        }
        acc = s[0] + s[1] + s[2] + s[3];
    }
    if (acc == 1234.5678f) out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int RA, int RB, int PRIO = 0>
static float run_share(float *out, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((share_kernel<RA, RB, PRIO>), dim3(256), dim3(512), 0, 0, out, iters, 0.001f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((share_kernel<RA, RB, PRIO>), dim3(256), dim3(512), 0, 0, out, iters, 0.001f);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5 * 1e3f;
}

// ---- split-K reduction variants --------------------------------------------------------------------------------------------
constexpr int kN = 10240;   // gradient slots per workgroup partial
__device__ __forceinline__ int xcc_id() {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15;
}
// mode 0: plain partial slab per WG (baseline)   1: f64 atomics agent scope, one slab   2: f64 atomics workgroup scope, slab[xcc]
// mode 3: f32 atomics agent, one slab            4: f32 workgroup scope slab[xcc]       5: f64 agent scope, slab[xcc]
template <int MODE>
__global__ void __launch_bounds__(512) reduce_kernel(float *slabs_f, double *slabs_d, int spin) {
    // some work first so that the WGs do not arrive in lock-step
    float v = (float)threadIdx.x;
    for (int i = 0; i < spin * (1 + (int)(blockIdx.x & 3)); ++i) v = fmaf(v, 1.0001f, 0.5f);
    const float val = 1.0f + (v == 123.f ? 1.f : 0.f);
    if (MODE == 0) {
        float *dst = slabs_f + (size_t)blockIdx.x * kN;
        for (int i = threadIdx.x; i < kN; i += 512) dst[i] = val;
    } else if (MODE == 1) {
        for (int i = threadIdx.x; i < kN; i += 512) __hip_atomic_fetch_add(slabs_d + i, (double)val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 2) {
        double *dst = slabs_d + (size_t)xcc_id() * kN;
        for (int i = threadIdx.x; i < kN; i += 512) __hip_atomic_fetch_add(dst + i, (double)val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (MODE == 3) {
        for (int i = threadIdx.x; i < kN; i += 512) __hip_atomic_fetch_add(slabs_f + i, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 4) {
        float *dst = slabs_f + (size_t)xcc_id() * kN;
        for (int i = threadIdx.x; i < kN; i += 512) __hip_atomic_fetch_add(dst + i, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (MODE == 5) {
        double *dst = slabs_d + (size_t)xcc_id() * kN;
        for (int i = threadIdx.x; i < kN; i += 512) __hip_atomic_fetch_add(dst + i, (double)val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int MODE>
static void run_reduce(float *sf, double *sd, int spin) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 10;
    CK(hipMemset(sf, 0, sizeof(float) * 256 * kN)); CK(hipMemset(sd, 0, sizeof(double) * 16 * kN));
    hipLaunchKernelGGL(reduce_kernel<MODE>, dim3(256), dim3(512), 0, 0, sf, sd, spin);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(reduce_kernel<MODE>, dim3(256), dim3(512), 0, 0, sf, sd, spin);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // correctness: totals over slabs must be 256 * (reps + 1) per slot
    std::vector<float> hf(16 * kN); std::vector<double> hd(16 * kN);
    CK(hipMemcpy(hf.data(), sf, sizeof(float) * 16 * kN, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hd.data(), sd, sizeof(double) * 16 * kN, hipMemcpyDeviceToHost));
    double bad = 0; int xccs = 0;
    if (MODE != 0) {
        for (int i = 0; i < kN; ++i) {
            double t = 0;
            for (int x = 0; x < 16; ++x) t += (MODE == 3 || MODE == 4) ? hf[(size_t)x * kN + i] : hd[(size_t)x * kN + i];
            if (t != 256.0 * (reps + 1)) bad += 1;
        }
        for (int x = 0; x < 16; ++x) xccs += ((MODE == 3 || MODE == 4) ? hf[(size_t)x * kN] : hd[(size_t)x * kN]) != 0;
    }
    printf("reduce mode %d spin %5d: %8.2f us per launch   wrong slots %.0f   slabs used %d\n", MODE, spin, ms / reps * 1e3f, bad, xccs);
}

int main() {
    float *out; CK(hipMalloc(&out, 256 * 512 * 4));
    const int it = 2000;   // 2000 * 16 MFMA * 32 cyc = 1.02M cycles ~ 0.43 ms at 2.4 GHz
    printf("share: MFMA alone            %8.1f us\n", run_share<1, 0>(out, it));
    printf("share: MFMA | MFMA           %8.1f us\n", run_share<1, 1>(out, it));
    printf("share: VALU alone (B)        %8.1f us\n", run_share<0, 2>(out, it));
    printf("share: MFMA | VALU           %8.1f us\n", run_share<1, 2>(out, it));
    printf("share: VALU | VALU           %8.1f us\n", run_share<2, 2>(out, it));
    printf("share: exp alone (B)         %8.1f us\n", run_share<0, 3>(out, it));
    printf("share: MFMA | exp            %8.1f us\n", run_share<1, 3>(out, it));
    printf("share: bpermute alone (B)    %8.1f us\n", run_share<0, 4>(out, it));
    printf("share: MFMA | bpermute       %8.1f us\n", run_share<1, 4>(out, it));
    printf("share: MFMA+VALU one stream  %8.1f us\n", run_share<5, 0>(out, it));
    printf("share: (MFMA+VALU) x2 waves  %8.1f us\n", run_share<5, 5>(out, it));
    printf("share: ds_read_b128 alone(B) %8.1f us\n", run_share<0, 6>(out, it));
    printf("share: MFMA | ds_read_b128   %8.1f us\n", run_share<1, 6>(out, it));
    printf("swap : bpermute(old) | MFMA(young)   %8.1f us\n", run_share<4, 1>(out, it));
    printf("swap : VALU(old) | MFMA(young)       %8.1f us\n", run_share<2, 1>(out, it));
    printf("prio : MFMA | bpermute@prio3         %8.1f us\n", run_share<1, 4, 1>(out, it));
    printf("prio : MFMA | VALU@prio3             %8.1f us\n", run_share<1, 2, 1>(out, it));
    printf("prio : MFMA | ds_read@prio3          %8.1f us\n", run_share<1, 6, 1>(out, it));
    printf("prio : MFMA@prio3 | bpermute         %8.1f us\n", run_share<1, 4, 2>(out, it));
    printf("swap : ds_read(old) | MFMA(young)    %8.1f us\n", run_share<6, 1>(out, it));
    float *sf; double *sd;
    CK(hipMalloc(&sf, sizeof(float) * 256 * kN)); CK(hipMalloc(&sd, sizeof(double) * 16 * kN));
    for (int spin : {0}) {
        run_reduce<0>(sf, sd, spin); run_reduce<1>(sf, sd, spin); run_reduce<2>(sf, sd, spin);
        run_reduce<3>(sf, sd, spin); run_reduce<4>(sf, sd, spin); run_reduce<5>(sf, sd, spin);
    }
    return 0;
}
