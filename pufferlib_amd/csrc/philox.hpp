// philox.hpp — Philox4x32-10 (Salmon, Moraes, Dror, Shaw; SC'11) as the counter-based source of the
// Exp(1) action noise that stands in for torch.multinomial's internal `exponential_` draw
// (SURVEY.md hard part 2).  Stream definition (ours; restated in oracle/puffer_oracle.c for checking):
//   key     = (seed & 0xffffffff, seed >> 32)
//   counter = (row, j, step & 0xffffffff, step >> 32)        row = global env index, j = action column / 4
//   words w0..w3 -> columns 4j..4j+3,  u = ((w >> 8) + 0.5) * 2^-24 in (0,1),  q = -log(u)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pfa {

struct u32x4 {
    uint32_t x, y, z, w;
};

__host__ __device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                        uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

__host__ __device__ __forceinline__ float philox_uniform(uint32_t w) {
    return ((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

}  // namespace pfa
