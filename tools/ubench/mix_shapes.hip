// Which MFMA shape overlaps better with the flow kernel's VALU mix on gfx950?
// Per "layer" of 32 queries: fp16 matrix work = 18 x 32x32x16  (or 36 x 16x16x32), VALU work = the
// realistic per-layer mix for 2 x 8 units per lane: per unit {exp, rcp, add, 5 plain, 3 and, 1.5 pk_sub, 3 cvt_pk}.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// one "unit" worth of VALU (about 15 instructions), independent across calls via the register set
#define UNIT(a, b, c, d)                                                             \
    asm volatile("v_exp_f32 %0, %0\n v_add_f32 %0, 1.0, %0\n v_rcp_f32 %0, %0\n"    \
                 "v_mul_f32 %1, %0, %1\n v_fma_f32 %2, %0, %1, %2\n v_fma_f32 %3, %1, %2, %0\n" \
                 "v_mul_f32 %1, %3, %1\n v_mul_f32 %2, %3, %2\n"                     \
                 "v_and_b32 %0, 0xffffe000, %1\n v_and_b32 %3, 0xffffe000, %2\n v_and_b32 %1, 0xffffe000, %1\n" \
                 "v_sub_f32 %2, %2, %3\n v_sub_f32 %1, %1, %0\n"                     \
                 "v_cvt_pk_f16_f32 %0, %0, %3\n v_cvt_pk_f16_f32 %1, %1, %2\n v_cvt_pk_f16_f32 %2, %2, %3\n" \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

template <int SHAPE, int ORDER>   // ORDER 0: all MFMA then all VALU ; 1: interleaved
__global__ __launch_bounds__(1024) void k(float* out, int iters, long long* cyc) {
    float a[16];
    for (int j = 0; j < 16; ++j) a[j] = threadIdx.x * 1e-3f + j * 0.01f;
    f32x4 c[6]; f32x16 d[3];
    for (int j = 0; j < 6; ++j) c[j] = (f32x4){a[0], a[1], a[2], a[3]};
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 16; ++i) d[j][i] = a[i];
    f16x8 h0, h1;
    for (int j = 0; j < 8; ++j) { h0[j] = (_Float16)(a[0] + j); h1[j] = (_Float16)(a[1] - j); }
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (ORDER == 0) {
            if (SHAPE == 0) {
#pragma unroll
                for (int u = 0; u < 36; ++u) c[u % 6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c[u % 6], 0, 0, 0);
            } else {
#pragma unroll
                for (int u = 0; u < 18; ++u) d[u % 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, h1, d[u % 3], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) UNIT(a[(4 * u) & 15], a[(4 * u + 1) & 15], a[(4 * u + 2) & 15], a[(4 * u + 3) & 15]);
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (SHAPE == 0) {
                    c[(2 * u) % 6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c[(2 * u) % 6], 0, 0, 0);
                    c[(2 * u + 1) % 6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c[(2 * u + 1) % 6], 0, 0, 0);
                    if (u < 4) c[(2 * u + 2) % 6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c[(2 * u + 2) % 6], 0, 0, 0);
                } else {
                    d[u % 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, h1, d[u % 3], 0, 0, 0);
                    if (u < 2) d[(u + 1) % 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, h1, d[(u + 1) % 3], 0, 0, 0);
                }
                UNIT(a[(4 * u) & 15], a[(4 * u + 1) & 15], a[(4 * u + 2) & 15], a[(4 * u + 3) & 15]);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 16; ++j) s += a[j];
    for (int j = 0; j < 6; ++j) s += c[j][0] + c[j][3];
    for (int j = 0; j < 3; ++j) s += d[j][0] + d[j][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicMax((unsigned long long*)cyc, (unsigned long long)(t1 - t0));
}

template <int SHAPE, int ORDER>
void run(const char* name) {
    static float* out = nullptr; static long long* cyc = nullptr;
    if (!out) { hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8); }
    for (int w : {1, 2, 3, 4}) {
        k<SHAPE, ORDER><<<256, 256 * w>>>(out, 10, cyc);
        hipDeviceSynchronize(); hipMemset(cyc, 0, 8);
        k<SHAPE, ORDER><<<256, 256 * w>>>(out, 500, cyc);
        hipDeviceSynchronize();
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-34s waves/SIMD=%d : %7.1f cycles per layer (32 queries) per SIMD\n", name, w, (double)c / 500 / w);
    }
}
int main() {
    run<0, 0>("16x16x32 x36, clustered");
    run<0, 1>("16x16x32 x36, interleaved");
    run<1, 0>("32x32x16 x18, clustered");
    run<1, 1>("32x32x16 x18, interleaved");
    return 0;
}
