// encoding.hip — the positional encoding as a stand-alone pass.
//
// Reference: `positional_encoding_1` (rendering/utils/model.py:9-57): out = cat([x] if include_input,
// sin(f_0 x), cos(f_0 x), ..., sin(f_{P-1} x), cos(f_{P-1} x)) on the last dimension, f = 2^linspace(0,
// P-1, P) (log_sampling) or linspace(1, 2^(P-1), P).  Inside the flow kernel the encoding is fused
// and never touches HBM; this un-fused form exists (a) as the drop-in for the reference function
// (training code and the velocity nets call it on [N,2] tensors) and (b) as the HBM-bound "encoding
// pass" whose GB/s BASELINE.json asks to be reported: 4 D bytes read, 4 D (1 + 2P) written per row
// (D = 2, P = 5: 8 + 88 = 96 B/row).
//
// A workgroup encodes a tile of 128 rows into LDS (one thread per (row, dim, band): ONE sincosf gives
// both the sin and the cos column) and then streams the tile out with fully coalesced 16-B stores —
// the tile's rows are contiguous in the output, whatever the row length.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "bsdfd.h"
#include "common.h"

namespace {

constexpr int ENC_TILE_ROWS = 128;
constexpr int ENC_MAX_BANDS = 16;
constexpr int ENC_MAX_DIM = 8;

struct EncParams {
    const float* x;
    float* out;
    long long n;
    int dim, bands, include_input, row_len;
    int tile_rows;  // rows per workgroup: 128, fewer for very long rows so that the tile stays within 60 KB of LDS
    float freq[ENC_MAX_BANDS];
};

// DIM / BANDS > 0: compile-time row shape (the hot path's [N,2] x 5 or 3 bands: no integer divisions,
// unrolled band loop); 0: run-time shape.
template <int DIM, int BANDS>
__global__ __launch_bounds__(256) void encode_kernel(const EncParams p) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int dim = DIM ? DIM : p.dim, bands = BANDS ? BANDS : p.bands;
    const int row_len = DIM ? DIM * (1 + 2 * BANDS) : p.row_len;  // the specialised shapes include the input
    const int base = (DIM || p.include_input) ? dim : 0;
    const long long row0 = (long long)blockIdx.x * p.tile_rows;
    const int rows = (int)min((long long)p.tile_rows, p.n - row0);
    // one thread per (row, dim): ONE sincosf per band gives the sin and the cos column
    for (int item = threadIdx.x; item < rows * dim; item += blockDim.x) {
        const int r = item / dim, d = item - r * dim;
        const float v = p.x[row0 * dim + item];
        float* o = tile + r * row_len;
        if (DIM || p.include_input) o[d] = v;
#pragma unroll
        for (int b = 0; b < bands; ++b) {
            float s, c;
            sincosf(v * (BANDS ? (float)(1 << b) : p.freq[b]), &s, &c);
            o[base + 2 * dim * b + d] = s;
            o[base + 2 * dim * b + dim + d] = c;
        }
    }
    __syncthreads();
    const long long out0 = row0 * row_len;
    const int total = rows * row_len;
    if (((out0 | total) & 3) == 0) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4* dst = reinterpret_cast<f32x4*>(p.out + out0);
        const f32x4* src = reinterpret_cast<const f32x4*>(tile);
        for (int i = threadIdx.x; i < total / 4; i += blockDim.x) __builtin_nontemporal_store(src[i], dst + i);  // write-once stream
    } else {
        for (int i = threadIdx.x; i < total; i += blockDim.x) p.out[out0 + i] = tile[i];
    }
}

}  // namespace

extern "C" int bsdfd_positional_encoding(const float* x, int64_t n, int32_t dim, int32_t bands, int32_t include_input,
                                         int32_t log_sampling, float* out, void* stream) {
    if (n < 0) return bsdfd_fail_(BSDFD_EINVAL, "N must be >= 0");
    if (dim < 1 || dim > ENC_MAX_DIM) return bsdfd_fail_(BSDFD_EINVAL, "dim must be in [1, 8]");
    if (bands < 0 || bands > ENC_MAX_BANDS) return bsdfd_fail_(BSDFD_EINVAL, "bands must be in [0, 16]");
    if (bands == 0 && !include_input) return bsdfd_fail_(BSDFD_EINVAL, "empty encoding");
    if (n == 0) return BSDFD_OK;
    if (!x || !out) return bsdfd_fail_(BSDFD_EINVAL, "null pointer");
    EncParams p;
    p.x = x; p.out = out; p.n = n; p.dim = dim; p.bands = bands; p.include_input = include_input ? 1 : 0;
    p.row_len = dim * (p.include_input + 2 * bands);
    for (int b = 0; b < ENC_MAX_BANDS; ++b) p.freq[b] = 0.0f;
    for (int b = 0; b < bands; ++b) {
        // torch.linspace(0, P-1, P) = 0, 1, ..., P-1 exactly; 2.0 ** k is exact.  Linear sampling:
        // linspace(1, 2^(P-1), P) evaluated as torch does (start + step * i, step in fp32).
        if (log_sampling) {
            p.freq[b] = (float)(1u << b);
        } else {
            const float hi = (float)(1u << (bands - 1));
            const float step = bands > 1 ? (hi - 1.0f) / (float)(bands - 1) : 0.0f;
            p.freq[b] = b < bands / 2 ? 1.0f + step * (float)b : hi - step * (float)(bands - 1 - b);
        }
    }
    p.tile_rows = ENC_TILE_ROWS;
    while (p.tile_rows > 1 && (size_t)p.tile_rows * p.row_len * sizeof(float) > 60 * 1024) p.tile_rows /= 2;
    const long long blocks = (n + p.tile_rows - 1) / p.tile_rows;
    if (blocks > 0x7fffffffLL) return bsdfd_fail_(BSDFD_EINVAL, "N too large for one launch");
    const size_t lds = (size_t)p.tile_rows * p.row_len * sizeof(float);
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dim == 2 && bands == 5 && include_input && log_sampling)       // velocity nets' conditioning
        hipLaunchKernelGGL((encode_kernel<2, 5>), grid, block, lds, st, p);
    else if (dim == 2 && bands == 3 && include_input && log_sampling)  // base-density nets
        hipLaunchKernelGGL((encode_kernel<2, 3>), grid, block, lds, st, p);
    else
        hipLaunchKernelGGL((encode_kernel<0, 0>), grid, block, lds, st, p);
    HIP_TRY(hipGetLastError());
    return BSDFD_OK;
}
