// episode_fin.hpp — EpisodeStats bookkeeping (pufferlib/postprocess.py:18-54) shared by the device-resident env families:
// per-env accumulators of the episodes finished since the last statistics read (what clean_pufferl.evaluate averages,
// clean_pufferl.py:127-137) and the episode finished by the last send() (what recv() hands out as `infos`).
#pragma once
#include "common.hpp"

namespace pfa {

struct EpisodeFin {
    double sum_return, sum_length, sum_score;
    double last_return, last_score;
    int finished, last_fin, last_length, pad;
};

__device__ __forceinline__ void episode_account(EpisodeFin &f, double ret, int len, double score) {
    f.sum_return += ret;
    f.sum_length += (double)len;
    f.sum_score += score;
    f.finished += 1;
    f.last_fin = 1;
    f.last_return = ret;
    f.last_length = len;
    f.last_score = score;
}

// out4 = {episodes finished, sum of returns, sum of lengths, sum of scores} since the last reset of the accumulators;
// single workgroup, fixed-order f64 sums (deterministic).
// `underrun` (optional): the env family's tape-underrun flag, reported as out[4] so that the trainer's one readback sees it.
static __global__ void __launch_bounds__(256) episode_stats_kernel(EpisodeFin *fin, int n, double *out4, int reset,
                                                                   const int *underrun = nullptr) {
    __shared__ double sh[4][256];
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int e = threadIdx.x; e < n; e += 256) {
        EpisodeFin &f = fin[e];
        a[0] += (double)f.finished;
        a[1] += f.sum_return;
        a[2] += f.sum_length;
        a[3] += f.sum_score;
        if (reset) {
            f.finished = 0;
            f.sum_return = f.sum_length = f.sum_score = 0.0;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] = a[k];
    __syncthreads();
    if (threadIdx.x < 4) {
        double s = 0.0;
        for (int i = 0; i < 256; ++i) s += sh[threadIdx.x][i];
        out4[threadIdx.x] = s;
    }
    if (threadIdx.x == 4 && underrun) out4[4] = (double)*underrun;
}

static __global__ void __launch_bounds__(256) episode_infos_kernel(const EpisodeFin *fin, int n, uint8_t *finished, double *ret, int *len,
                                                           double *score) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const EpisodeFin &f = fin[e];
    finished[e] = f.last_fin ? 1 : 0;
    ret[e] = f.last_return;
    len[e] = f.last_length;
    score[e] = f.last_score;
}

}  // namespace pfa
