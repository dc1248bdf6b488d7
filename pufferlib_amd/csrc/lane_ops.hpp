// lane_ops.hpp — reductions over the 16 lanes of a DPP row with data-parallel-primitive moves (one VALU op each)
// instead of ds_bpermute round trips through the LDS pipe: quad_perm[1,0,3,2], quad_perm[2,3,0,1], row_half_mirror,
// row_mirror.  After the four steps every lane of the row holds the reduction of all 16.
#pragma once
#include "common.hpp"

namespace pfa {

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int x) {
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, false);
}

constexpr int kDppXor1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // lane i <-> 7-i within each 8
constexpr int kDppMirror = 0x140;      // lane i <-> 15-i within the row

__device__ __forceinline__ float row16_max(float x) {
    x = fmaxf(x, dpp_f<kDppXor1>(x));
    x = fmaxf(x, dpp_f<kDppXor2>(x));
    x = fmaxf(x, dpp_f<kDppHalfMirror>(x));
    x = fmaxf(x, dpp_f<kDppMirror>(x));
    return x;
}

// Fixed combination order -> the same bits on every lane and in every kernel that uses it.
__device__ __forceinline__ float row16_sum(float x) {
    x = x + dpp_f<kDppXor1>(x);
    x = x + dpp_f<kDppXor2>(x);
    x = x + dpp_f<kDppHalfMirror>(x);
    x = x + dpp_f<kDppMirror>(x);
    return x;
}

// argmax with the lowest index winning ties (torch.argmax's rule on CPU)
__device__ __forceinline__ void row16_argmax(float &best, int &idx) {
#define PFA_ARGMAX_STEP(CTRL)                                   \
    {                                                           \
        const float ob = dpp_f<CTRL>(best);                     \
        const int oi = dpp_i<CTRL>(idx);                        \
        const bool take = (ob > best) | ((ob == best) & (oi < idx)); /* bitwise: no exec-mask branches */ \
        best = take ? ob : best;                                \
        idx = take ? oi : idx;                                  \
    }
    PFA_ARGMAX_STEP(kDppXor1)
    PFA_ARGMAX_STEP(kDppXor2)
    PFA_ARGMAX_STEP(kDppHalfMirror)
    PFA_ARGMAX_STEP(kDppMirror)
#undef PFA_ARGMAX_STEP
}

}  // namespace pfa
