// nativize.hip — structured-observation unpack: the device side of pufferlib.pytorch.nativize_tensor
// (pufferlib/pytorch.py:96-145).  The reference hands policies flat emulated rows [N, D] (bytes when the leaves of the Dict /
// Tuple observation space have mixed dtypes, emulation.py:96-110) and reinterprets column ranges as typed sub-tensors with
// narrow().view(dtype).view(N, *shape) — strided views that every consumer then gathers from, usually right after a
// .float().  Here ONE launch turns the array of structs into one dense, aligned tensor per leaf (or into column ranges of
// one [N, total] f32 matrix: the torch.cat(...).float() most encoders start with), reading every input byte once.
//
// HBM-bound byte work, no arithmetic worth naming: algorithmic traffic = N*row_bytes in + N*sum(leaf bytes) out.  Rows up
// to 4 KB: a workgroup stages a tile of whole rows in LDS with 16-byte loads (tile start is 16-byte aligned because the
// tile holds a multiple of 16 rows) and writes each leaf's slab of the tile — which is contiguous in the output — with
// dword stores; LDS gathers absorb the arbitrary field offsets.  Longer rows: leaf slabs are long enough to copy directly.
// Measured on MI355X (4M rows of 108-112 bytes): 5.2-5.5 TB/s for dense leaves (the rate of a plain device-to-device copy
// here, 5.2-5.4 TB/s), 5.1-5.5 TB/s with the fused .float(), 4.4-4.9 TB/s for the fused f32 matrix.  Tiles of 16 KB: 8 KB
// and 32-48 KB tiles were 5-25 % slower (fewer workgroups in flight per CU).
#include <hip/hip_fp16.h>

#include "common.hpp"

namespace pfa {

constexpr int kNatThreads = 256;
constexpr int kNatTileBytes = 16 * 1024;        // input rows per workgroup: small tiles keep 8+ workgroups per CU in flight
constexpr int kNatMatrixTileBytes = 20 * 1024;  // input rows + their f32 matrix rows (matrix mode); 12/28/40 KB measured slower
constexpr int kNatRowsAlign = 16;
constexpr int kNatMaxRowBytesTiled = 4096;  // 16 rows * 4096 B = 64 KB tile at most

struct NatField {
    void *out;          // leaf tensor (raw mode) or the first column of this leaf in the f32 matrix
    int32_t offset;     // byte offset of the leaf inside a row
    int32_t count;      // elements per row
    int32_t code;       // pfa_nat_dtype of the source elements
    int32_t to_f32;     // 0: copy elements as they are, 1: convert to f32
    int32_t out_stride; // elements between rows of `out` (== count for a dense leaf tensor)
    int32_t isz;        // source element size in bytes
    uint32_t magic;     // ceil(2^32 / d) for the per-row divisor d of the fast loops (bytes per row or elements per row)
    int32_t pad;
};

struct NatArgs {
    NatField f[PFA_NAT_MAX_FIELDS];
    const uint8_t *rows;
    long long n;
    int32_t row_bytes;
    int32_t nfields;
    int32_t tile_rows;
    int32_t matrix_cols;  // > 0: every leaf is a column range of ONE dense [n][matrix_cols] f32 matrix starting at f[0].out
};

__host__ __device__ inline int nat_itemsize(int code) {
    switch (code) {
        case PFA_NAT_U8: case PFA_NAT_I8: return 1;
        case PFA_NAT_U16: case PFA_NAT_I16: case PFA_NAT_F16: return 2;
        case PFA_NAT_U32: case PFA_NAT_I32: case PFA_NAT_F32: return 4;
        case PFA_NAT_U64: case PFA_NAT_I64: case PFA_NAT_F64: return 8;
    }
    return 0;
}

// b / d for b * d < 2^32 (tile bytes < 2^16 and d < 2^16): one mul_hi
__device__ __forceinline__ uint32_t fast_div(uint32_t b, uint32_t magic, uint32_t d) { return d == 1 ? b : __umulhi(b, magic); }

template <typename Src>
__device__ __forceinline__ uint64_t gather_bytes(const Src *p, int nbytes) {  // little endian, any alignment
    uint64_t v = 0;
    for (int i = 0; i < nbytes; ++i) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

// element bits from the LDS tile: one typed read when the element is naturally aligned there, bytes otherwise
__device__ __forceinline__ uint64_t load_bits(const uint8_t *p, int isz, bool natural) {
    if (!natural) return gather_bytes(p, isz);
    switch (isz) {
        case 1: return *p;
        case 2: return *reinterpret_cast<const uint16_t *>(p);
        case 4: return *reinterpret_cast<const uint32_t *>(p);
        default: return *reinterpret_cast<const uint64_t *>(p);
    }
}

__device__ __forceinline__ float nat_to_f32(uint64_t bits, int code) {
    switch (code) {
        case PFA_NAT_U8: return (float)(uint8_t)bits;
        case PFA_NAT_I8: return (float)(int8_t)bits;
        case PFA_NAT_U16: return (float)(uint16_t)bits;
        case PFA_NAT_I16: return (float)(int16_t)bits;
        case PFA_NAT_U32: return (float)(uint32_t)bits;
        case PFA_NAT_I32: return (float)(int32_t)bits;
        case PFA_NAT_U64: return (float)bits;
        case PFA_NAT_I64: return (float)(int64_t)bits;
        case PFA_NAT_F16: {
            const uint16_t h = (uint16_t)bits;
            return __half2float(*reinterpret_cast<const __half *>(&h));
        }
        case PFA_NAT_F32: return __uint_as_float((uint32_t)bits);
        case PFA_NAT_F64: return (float)__longlong_as_double((long long)bits);
    }
    return 0.0f;
}

// ---- rows of up to 4 KB: LDS tile of whole rows -------------------------------------------------------------
__global__ void __launch_bounds__(kNatThreads) nativize_tiled_kernel(NatArgs a) {
    extern __shared__ __align__(16) uint8_t tile[];
    const long long r0 = (long long)blockIdx.x * a.tile_rows;
    const int nr = (int)min((long long)a.tile_rows, a.n - r0);
    const int D = a.row_bytes;
    const uint32_t total = (uint32_t)nr * (uint32_t)D;
    const uint8_t *src = a.rows + r0 * D;  // 16-byte aligned: r0 is a multiple of 16 rows
    const uint32_t nvec = total >> 4;
    for (uint32_t i = threadIdx.x; i < nvec; i += kNatThreads)
        reinterpret_cast<uint4 *>(tile)[i] = reinterpret_cast<const uint4 *>(src)[i];
    for (uint32_t i = (nvec << 4) + threadIdx.x; i < total; i += kNatThreads) tile[i] = src[i];
    __syncthreads();

    if (a.matrix_cols > 0) {
        // torch.cat(leaves).float(): the tile's rows of the matrix are one contiguous run of floats.  Leaves are converted
        // into an LDS image of that run (leaf by leaf: uniform dtype per loop), which then leaves with 16-byte stores — row
        // segments of a few floats written straight to HBM cost a third of the bandwidth.
        float *otile = reinterpret_cast<float *>(tile + (((size_t)a.tile_rows * D + 15) & ~(size_t)15));
        const uint32_t cols = (uint32_t)a.matrix_cols;
        for (int fi = 0; fi < a.nfields; ++fi) {
            const NatField f = a.f[fi];
            const uint32_t col0 = (uint32_t)(reinterpret_cast<float *>(f.out) - reinterpret_cast<float *>(a.f[0].out));
            const uint32_t cnt = (uint32_t)f.count, ne = (uint32_t)nr * cnt;
            const bool natural = ((uint32_t)(f.offset | D) & (uint32_t)(f.isz - 1)) == 0;
            for (uint32_t e = threadIdx.x; e < ne; e += kNatThreads) {
                const uint32_t r = fast_div(e, f.magic, cnt), k = e - r * cnt;
                otile[r * cols + col0 + k] = nat_to_f32(load_bits(tile + r * D + f.offset + k * f.isz, f.isz, natural), f.code);
            }
        }
        __syncthreads();
        float *out = reinterpret_cast<float *>(a.f[0].out) + r0 * cols;  // 16-byte aligned: r0 is a multiple of 16 rows
        const uint32_t nf = (uint32_t)nr * cols, nv = nf >> 2;
        for (uint32_t i = threadIdx.x; i < nv; i += kNatThreads) reinterpret_cast<float4 *>(out)[i] = reinterpret_cast<const float4 *>(otile)[i];
        for (uint32_t i = (nv << 2) + threadIdx.x; i < nf; i += kNatThreads) out[i] = otile[i];
        return;
    }
    for (int fi = 0; fi < a.nfields; ++fi) {
        const NatField f = a.f[fi];
        if (f.to_f32) {
            float *out = reinterpret_cast<float *>(f.out) + r0 * f.out_stride;
            const uint32_t cnt = (uint32_t)f.count, ne = (uint32_t)nr * cnt;
            const bool natural = ((uint32_t)(f.offset | D) & (uint32_t)(f.isz - 1)) == 0;
            for (uint32_t e = threadIdx.x; e < ne; e += kNatThreads) {
                const uint32_t r = fast_div(e, f.magic, cnt), k = e - r * cnt;
                const uint8_t *p = tile + r * D + f.offset + k * f.isz;
                out[(size_t)r * f.out_stride + k] = nat_to_f32(load_bits(p, f.isz, natural), f.code);
            }
        } else {  // the tile's slab of this leaf is contiguous in the output and starts 16-byte aligned
            const uint32_t fb = (uint32_t)f.count * (uint32_t)f.isz, nb = (uint32_t)nr * fb;
            uint8_t *out = reinterpret_cast<uint8_t *>(f.out) + r0 * fb;
            const bool aligned = ((fb | (uint32_t)f.offset | (uint32_t)D) & 3u) == 0;
            const uint32_t nd = nb >> 2;
            for (uint32_t j = threadIdx.x; j < nd; j += kNatThreads) {
                const uint32_t b = j << 2;
                uint32_t w;
                if (aligned) {
                    const uint32_t r = fast_div(b, f.magic, fb), o = b - r * fb;
                    w = *reinterpret_cast<const uint32_t *>(tile + r * D + f.offset + o);
                } else {
                    w = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t bi = b + i, r = fast_div(bi, f.magic, fb), o = bi - r * fb;
                        w |= (uint32_t)tile[r * D + f.offset + o] << (8 * i);
                    }
                }
                reinterpret_cast<uint32_t *>(out)[j] = w;
            }
            for (uint32_t b = (nd << 2) + threadIdx.x; b < nb; b += kNatThreads) {
                const uint32_t r = fast_div(b, f.magic, fb), o = b - r * fb;
                out[b] = tile[r * D + f.offset + o];
            }
        }
    }
}

// ---- long rows: one workgroup per (row block, leaf); the leaf's bytes of a row are a long contiguous run ------
__global__ void __launch_bounds__(kNatThreads) nativize_direct_kernel(NatArgs a, int rows_per_block) {
    const NatField f = a.f[blockIdx.y];
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const int nr = (int)min((long long)rows_per_block, a.n - r0);
    const size_t D = (size_t)a.row_bytes;
    for (int r = 0; r < nr; ++r) {
        const uint8_t *src = a.rows + (size_t)(r0 + r) * D + f.offset;
        if (f.to_f32) {
            float *out = reinterpret_cast<float *>(f.out) + (size_t)(r0 + r) * f.out_stride;
            const bool natural = (((uintptr_t)src) & (uintptr_t)(f.isz - 1)) == 0;
            // four elements per lane (one 4..16-byte load, one 16-byte store) where the run allows it
            const int chunk = 4 * f.isz;
            const bool vec = f.isz <= 4 && (((uintptr_t)src) & (uintptr_t)(chunk - 1)) == 0 && (((uintptr_t)out) & 15) == 0;
            const int nv = vec ? f.count >> 2 : 0;
            for (int q = threadIdx.x; q < nv; q += kNatThreads) {
                const uint8_t *p = src + (size_t)q * chunk;
                float4 v;
                if (f.isz == 1) {
                    const uint32_t w = *reinterpret_cast<const uint32_t *>(p);
                    v = make_float4(nat_to_f32(w & 0xff, f.code), nat_to_f32((w >> 8) & 0xff, f.code), nat_to_f32((w >> 16) & 0xff, f.code),
                                    nat_to_f32(w >> 24, f.code));
                } else if (f.isz == 2) {
                    const uint2 w = *reinterpret_cast<const uint2 *>(p);
                    v = make_float4(nat_to_f32(w.x & 0xffff, f.code), nat_to_f32(w.x >> 16, f.code), nat_to_f32(w.y & 0xffff, f.code),
                                    nat_to_f32(w.y >> 16, f.code));
                } else {
                    const uint4 w = *reinterpret_cast<const uint4 *>(p);
                    v = make_float4(nat_to_f32(w.x, f.code), nat_to_f32(w.y, f.code), nat_to_f32(w.z, f.code), nat_to_f32(w.w, f.code));
                }
                reinterpret_cast<float4 *>(out)[q] = v;
            }
            for (int k = (nv << 2) + threadIdx.x; k < f.count; k += kNatThreads)
                out[k] = nat_to_f32(load_bits(src + (size_t)k * f.isz, f.isz, natural), f.code);
        } else {
            const size_t fb = (size_t)f.count * f.isz;
            uint8_t *out = reinterpret_cast<uint8_t *>(f.out) + (size_t)(r0 + r) * fb;
            if ((((uintptr_t)src | (uintptr_t)out | fb) & 15) == 0) {
                for (size_t i = threadIdx.x; i < (fb >> 4); i += kNatThreads)
                    reinterpret_cast<uint4 *>(out)[i] = reinterpret_cast<const uint4 *>(src)[i];
            } else if ((((uintptr_t)src | (uintptr_t)out | fb) & 3) == 0) {
                for (size_t i = threadIdx.x; i < (fb >> 2); i += kNatThreads)
                    reinterpret_cast<uint32_t *>(out)[i] = reinterpret_cast<const uint32_t *>(src)[i];
            } else {
                for (size_t i = threadIdx.x; i < fb; i += kNatThreads) out[i] = src[i];
            }
        }
    }
}

}  // namespace pfa

using namespace pfa;

extern "C" int pfa_nativize_rows(const void *rows, int64_t num_rows, int32_t row_bytes, const pfa_nat_field *fields, int32_t num_fields,
                                 pfa_stream_t stream) {
    PFA_REQUIRE(rows && fields && num_rows >= 0 && row_bytes >= 1 && num_fields >= 1 && num_fields <= PFA_NAT_MAX_FIELDS,
                "nativize: bad arguments (1..%d fields)", PFA_NAT_MAX_FIELDS);
    PFA_REQUIRE(((uintptr_t)rows & 15) == 0, "nativize: rows must be 16-byte aligned");
    NatArgs a = {};
    a.rows = (const uint8_t *)rows;
    a.n = num_rows;
    a.row_bytes = row_bytes;
    a.nfields = num_fields;
    const bool tiled = row_bytes <= kNatMaxRowBytesTiled;
    for (int i = 0; i < num_fields; ++i) {
        const pfa_nat_field &s = fields[i];
        const int isz = nat_itemsize(s.dtype);
        PFA_REQUIRE(isz != 0, "nativize: field %d has unknown dtype code %d", i, (int)s.dtype);
        PFA_REQUIRE(s.out && s.count >= 1 && s.offset >= 0 && (int64_t)s.offset + (int64_t)s.count * isz <= row_bytes,
                    "nativize: field %d (offset %d, %d elements of %d bytes) does not fit rows of %d bytes", i, (int)s.offset,
                    (int)s.count, isz, (int)row_bytes);
        PFA_REQUIRE(s.to_f32 ? (s.out_stride >= s.count && ((uintptr_t)s.out & 3) == 0) : (s.out_stride == s.count && ((uintptr_t)s.out & 15) == 0),
                    "nativize: field %d: a raw leaf must be dense and 16-byte aligned, an f32 leaf 4-byte aligned with stride >= count", i);
        NatField &f = a.f[i];
        f.out = s.out;
        f.offset = s.offset;
        f.count = s.count;
        f.code = s.dtype;
        f.to_f32 = s.to_f32 ? 1 : 0;
        f.out_stride = s.out_stride;
        f.isz = isz;
        const uint32_t d = s.to_f32 ? (uint32_t)s.count : (uint32_t)s.count * (uint32_t)isz;
        f.magic = d > 1 ? (uint32_t)((0x100000000ull + d - 1) / d) : 0;
    }
    if (num_rows == 0) return 0;
    ScopedKernelTimer timer("nativize", (hipStream_t)stream);
    if (tiled) {
        // one dense f32 matrix?  (every leaf converted, one row stride == the sum of the counts, column ranges back to back)
        int64_t cols = 0;
        bool matrix = ((uintptr_t)a.f[0].out & 15) == 0;
        for (int i = 0; i < num_fields && matrix; ++i) {
            matrix = a.f[i].to_f32 && a.f[i].out == (void *)((float *)a.f[0].out + cols);
            cols += a.f[i].count;
        }
        for (int i = 0; i < num_fields && matrix; ++i) matrix = a.f[i].out_stride == cols;
        const int64_t per_row = row_bytes + (matrix ? cols * 4 : 0);
        if (matrix && (per_row * kNatRowsAlign > 64 * 1024 || cols * kNatRowsAlign >= 65536)) matrix = false;
        const int64_t tile_bytes = matrix ? kNatMatrixTileBytes : kNatTileBytes;
        int tile_rows = (int)((tile_bytes / (matrix ? per_row : row_bytes)) / kNatRowsAlign * kNatRowsAlign);
        if (tile_rows < kNatRowsAlign) tile_rows = kNatRowsAlign;
        if (tile_rows > 4096) tile_rows = 4096;
        while (matrix && (int64_t)tile_rows * cols >= 65536) tile_rows -= kNatRowsAlign;  // fast_div range of the element loops
        a.tile_rows = tile_rows;
        a.matrix_cols = matrix ? (int32_t)cols : 0;
        const size_t lds = align_up((size_t)tile_rows * row_bytes, 16) + (matrix ? (size_t)tile_rows * cols * 4 : 0);
        const long long blocks = (num_rows + tile_rows - 1) / tile_rows;
        PFA_REQUIRE(blocks < (1ll << 31), "nativize: too many rows");
        if (lds > 48 * 1024) {
            PFA_CHECK_HIP(hipFuncSetAttribute((const void *)nativize_tiled_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        }
        hipLaunchKernelGGL(nativize_tiled_kernel, dim3((unsigned)blocks), dim3(kNatThreads), lds, (hipStream_t)stream, a);
    } else {
        const int rows_per_block = 4;
        const long long blocks = (num_rows + rows_per_block - 1) / rows_per_block;
        PFA_REQUIRE(blocks < (1ll << 31), "nativize: too many rows");
        hipLaunchKernelGGL(nativize_direct_kernel, dim3((unsigned)blocks, (unsigned)num_fields), dim3(kNatThreads), 0, (hipStream_t)stream, a,
                           rows_per_block);
    }
    PFA_LAUNCH_CHECK();
    return 0;
}
