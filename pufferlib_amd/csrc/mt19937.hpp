// mt19937.hpp — MT19937 (Matsumoto & Nishimura) pieces needed to reproduce CPython's `random` stream
// on device: init_by_array seeding as done by random.seed(int) (CPython Modules/_randommodule.c),
// the 624-word regeneration and the output tempering.  The reference consumes this stream through
// random.sample at pufferlib/environments/ocean/ocean.py:449-459.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pfa {

constexpr int kMtN = 624;
constexpr int kMtM = 397;

__host__ __device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

__host__ __device__ __forceinline__ uint32_t mt_twist(uint32_t cur, uint32_t nxt, uint32_t far) {
    const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// One generator per thread, words strided by `stride` (word-major scratch so that lanes coalesce).
// random.seed(abs_seed): init_genrand(19650218) then init_by_array(32-bit digits of abs_seed).
__device__ inline void mt_seed_strided(uint32_t *mt, size_t stride, uint64_t abs_seed) {
    uint32_t key[2] = {(uint32_t)(abs_seed & 0xffffffffu), (uint32_t)(abs_seed >> 32)};
    const int key_len = key[1] ? 2 : 1;
    uint32_t prev = 19650218u;
    mt[0] = prev;
    for (int i = 1; i < kMtN; ++i) {
        prev = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)i;
        mt[(size_t)i * stride] = prev;
    }
    int i = 1, j = 0;
    prev = mt[0];
    for (int k = kMtN; k; --k) {  // kMtN > key_len always
        uint32_t v = (mt[(size_t)i * stride] ^ ((prev ^ (prev >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        mt[(size_t)i * stride] = v;
        prev = v;
        ++i; ++j;
        if (i >= kMtN) { mt[0] = prev; i = 1; }
        if (j >= key_len) j = 0;
    }
    for (int k = kMtN - 1; k; --k) {
        uint32_t v = (mt[(size_t)i * stride] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)i;
        mt[(size_t)i * stride] = v;
        prev = v;
        ++i;
        if (i >= kMtN) { mt[0] = prev; i = 1; }
    }
    mt[0] = 0x80000000u;
}

// Sequential in-place regeneration of one strided state (used once per env at async_reset).
__device__ inline void mt_regenerate_strided(uint32_t *mt, size_t stride) {
    const uint32_t first = mt[0];
    uint32_t cur = first;
    for (int kk = 0; kk < kMtN; ++kk) {
        const uint32_t nxt = (kk + 1 < kMtN) ? mt[(size_t)(kk + 1) * stride] : mt[0];
        const int f = kk + kMtM < kMtN ? kk + kMtM : kk + kMtM - kMtN;
        const uint32_t v = mt_twist(cur, nxt, mt[(size_t)f * stride]);
        mt[(size_t)kk * stride] = v;
        cur = nxt;
    }
}

// Workgroup-parallel regeneration (>= 624 threads): three dependency waves separated by barriers.
// new_[.] = next MT block computed from old_[.]; out[.] = tempered new_.  All threads; ends with a barrier.
__device__ __forceinline__ void mt_next_block(const uint32_t *old_, uint32_t *new_, uint32_t *out) {
    const int t = threadIdx.x;
    if (t < 227) {  // kk in [0,227): old words only
        const uint32_t x = mt_twist(old_[t], old_[t + 1], old_[t + kMtM]);
        new_[t] = x;
        out[t] = mt_temper(x);
    }
    __syncthreads();
    if (t >= 227 && t < 454) {  // kk in [227,454): new [0,227)
        const uint32_t x = mt_twist(old_[t], old_[t + 1], new_[t - 227]);
        new_[t] = x;
        out[t] = mt_temper(x);
    }
    __syncthreads();
    if (t >= 454 && t < kMtN) {  // kk in [454,624): new [227,397); kk = 623 wraps to new[0]
        const uint32_t x = mt_twist(old_[t], t + 1 < kMtN ? old_[t + 1] : new_[0], new_[t - 227]);
        new_[t] = x;
        out[t] = mt_temper(x);
    }
    __syncthreads();
}


}  // namespace pfa
