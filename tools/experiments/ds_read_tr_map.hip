// ds_read_tr_map.hip — EXPERIMENT: which LDS elements ds_read_b64_tr_b16 hands to which lane (gfx950), for the next round's transposed
// bf16 operands.  LDS holds 16-bit values equal to their own element index; every lane passes the address of "its" 8-byte row piece
// in two arrangements; the program prints what each lane received.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/ds_read_tr_map.hip -o tools/experiments/_bin/ds_read_tr_map
// RESULT (MI355X, ROCm 7.2): within each group of 16 lanes, with in[j][b] = element b (0..3) of the 8 bytes lane j points at,
//     lane L = 4 a + b receives  in[a][b], in[a + 4][b], in[a + 8][b], in[a + 12][b]           (out[a][b][e] = in[4 e + a][b])
// i.e. the high two bits of the source lane become the element index and the element index becomes the low two lane bits.  Intended use:
// an MFMA operand whose contraction index is the ROW of a row-major LDS tile — lane j points at (row 4 g + j / 4, columns 4 (j % 4) .. + 3)
// of a 4 x 16 block and lane n = 4 a + b ends up with rows 4 g .. 4 g + 3 of column n: the four k-values of column n the MFMA wants from
// it (two such reads for the eight k-values of v_mfma_f32_16x16x32_bf16).  No transposing loader, no VALU.
#include <hip/hip_runtime.h>

#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void k(short *out, int stride_elems, int mode) {
    __shared__ short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    // mode 0: lane L points at row L (row stride `stride_elems`), columns 0..3
    // mode 1: lane (c = L % 16, g = L / 16) points at row c, columns 4g..4g+3
    const int off = mode == 0 ? lane * stride_elems : (lane & 15) * stride_elems + 4 * (lane >> 4);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds + off));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}

int main() {
    short *d, h[256];
    hipMalloc(&d, sizeof(h));
    for (int mode = 0; mode < 2; ++mode) {
        const int stride = 64;     // elements per LDS row: element index = 64 * row + column
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (element = 64 * row + column):\n", mode);
        for (int lane = 0; lane < 64; ++lane) {
            printf("  lane %2d:", lane);
            for (int e = 0; e < 4; ++e) printf(" (r%2d,c%2d)", h[lane * 4 + e] / 64, h[lane * 4 + e] % 64);
            printf("%s", lane % 2 ? "\n" : "   ");
        }
    }
    return 0;
}
