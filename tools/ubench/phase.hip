// Do co-resident waves overlap MFMA bursts with VALU bursts when each wave's work is PHASE-SEPARATED
// (a burst of 18 MFMAs, then a burst of ~128 VALU), and does staggering the waves help?
// All instructions are inline asm so the compiler cannot re-interleave them.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define MFMA6(c0, c1, c2, c3, c4, c5)                                                       \
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %6, %7, %0\n v_mfma_f32_16x16x32_f16 %1, %6, %7, %1\n" \
                 "v_mfma_f32_16x16x32_f16 %2, %6, %7, %2\n v_mfma_f32_16x16x32_f16 %3, %6, %7, %3\n" \
                 "v_mfma_f32_16x16x32_f16 %4, %6, %7, %4\n v_mfma_f32_16x16x32_f16 %5, %6, %7, %5\n" \
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5) : "v"(h0), "v"(h1))
#define UNIT(a, b, c, d)                                                             \
    asm volatile("v_exp_f32 %0, %0\n v_add_f32 %0, 1.0, %0\n v_rcp_f32 %0, %0\n"    \
                 "v_mul_f32 %1, %0, %1\n v_fma_f32 %2, %0, %1, %2\n v_fma_f32 %3, %1, %2, %0\n" \
                 "v_mul_f32 %1, %3, %1\n v_mul_f32 %2, %3, %2\n"                     \
                 "v_and_b32 %0, 0xffffe000, %1\n v_and_b32 %3, 0xffffe000, %2\n v_and_b32 %1, 0xffffe000, %1\n" \
                 "v_sub_f32 %2, %2, %3\n v_sub_f32 %1, %1, %0\n"                     \
                 "v_cvt_pk_f16_f32 %0, %0, %3\n v_cvt_pk_f16_f32 %1, %1, %2\n v_cvt_pk_f16_f32 %2, %2, %3\n" \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

template <int MODE>  // 0: burst MFMA then burst VALU; 1: same + per-wave initial stagger; 2: fine interleave (2 MFMA + 1 UNIT) x 9
__global__ __launch_bounds__(1024) void k(float* out, int iters, long long* cyc) {
    float a[16];
    for (int j = 0; j < 16; ++j) a[j] = threadIdx.x * 1e-3f + j * 0.01f;
    f32x4 c[6];
    for (int j = 0; j < 6; ++j) c[j] = (f32x4){a[0], a[1], a[2], a[3]};
    f16x8 h0, h1;
    for (int j = 0; j < 8; ++j) { h0[j] = (_Float16)(a[0] + j); h1[j] = (_Float16)(a[1] - j); }
    const int wave_in_simd = threadIdx.x >> 8;  // waves 0-3 -> SIMD 0-3 first, then the next wave per SIMD
    long long t0 = __builtin_readcyclecounter();
    if (MODE == 1) {  // stagger: wave k of a SIMD starts with k * (1/waves) of a period of pure VALU
        for (int s = 0; s < wave_in_simd * 3; ++s) { UNIT(a[0], a[1], a[2], a[3]); UNIT(a[4], a[5], a[6], a[7]); }
    }
    for (int i = 0; i < iters; ++i) {
        if (MODE != 2) {
            MFMA6(c[0], c[1], c[2], c[3], c[4], c[5]);
            MFMA6(c[0], c[1], c[2], c[3], c[4], c[5]);
            MFMA6(c[0], c[1], c[2], c[3], c[4], c[5]);
#pragma unroll
            for (int u = 0; u < 8; ++u) UNIT(a[(4 * u) & 15], a[(4 * u + 1) & 15], a[(4 * u + 2) & 15], a[(4 * u + 3) & 15]);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_f16 %1, %2, %3, %1\n"
                             : "+v"(c[(2 * u) % 6]), "+v"(c[(2 * u + 1) % 6]) : "v"(h0), "v"(h1));
                UNIT(a[(4 * u) & 15], a[(4 * u + 1) & 15], a[(4 * u + 2) & 15], a[(4 * u + 3) & 15]);
            }
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_f16 %1, %2, %3, %1\n"
                         : "+v"(c[4]), "+v"(c[5]) : "v"(h0), "v"(h1));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 16; ++j) s += a[j];
    for (int j = 0; j < 6; ++j) s += c[j][0] + c[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicMax((unsigned long long*)cyc, (unsigned long long)(t1 - t0));
}

template <int MODE>
void run(const char* name) {
    static float* out = nullptr; static long long* cyc = nullptr;
    if (!out) { hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8); }
    for (int w : {1, 2, 3, 4}) {
        k<MODE><<<256, 256 * w>>>(out, 10, cyc);
        hipDeviceSynchronize(); hipMemset(cyc, 0, 8);
        k<MODE><<<256, 256 * w>>>(out, 1000, cyc);
        hipDeviceSynchronize();
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-40s waves/SIMD=%d : %7.1f cycles per layer-tile (18 MFMA + 8 units) per SIMD\n", name, w, (double)c / 1000 / w);
    }
}
int main() {
    run<0>("burst MFMA / burst VALU");
    run<1>("burst + staggered start");
    run<2>("fine interleave (2 MFMA : 1 unit)");
    return 0;
}
