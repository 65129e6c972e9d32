// Micro-benchmark: how many VALU instructions hide behind one MFMA on gfx950 (same wave and
// across co-resident waves of a SIMD).  cycles per (MFMA + K fillers) group, real shader cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int FILL>
__device__ __forceinline__ void filler(float& a, float& b) {
    if (FILL == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a));
    if (FILL == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
    if (FILL == 2) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (FILL == 3) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(*(double*)&a));
}

template <int K, int FILL, int SHAPE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, long long* cyc) {
    float a[8] __attribute__((aligned(8)));
    for (int j = 0; j < 8; ++j) a[j] = threadIdx.x * 1e-3f + j;
    f32x4 c[4]; f32x16 d[2];
    for (int j = 0; j < 4; ++j) c[j] = (f32x4){a[0], a[1], a[2], a[3]};
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) d[j][i] = a[i & 7];
    f16x8 h0, h1;
    for (int j = 0; j < 8; ++j) { h0[j] = (_Float16)(a[0] + j); h1[j] = (_Float16)(a[1] - j); }
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (SHAPE == 0) c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c[u & 3], 0, 0, 0);
            if (SHAPE == 1) d[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, h1, d[u & 1], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < K; ++f) filler<FILL>(a[(2 * (u * K + f)) & 6], a[((2 * (u * K + f)) & 6) + 1]);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 8; ++j) s += a[j];
    for (int j = 0; j < 4; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) s += d[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    // the slowest wave of block 0 (all of its waves start together on one CU): under contention the
    // oldest wave is favoured and finishes early, so wave 0's own time would under-report
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicMax((unsigned long long*)cyc, (unsigned long long)(t1 - t0));
}

template <int K, int FILL, int SHAPE>
double run(int waves_per_simd) {
    static float* out = nullptr; static long long* cyc = nullptr;
    if (!out) { hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8); }
    const int iters = 1000, threads = 256 * waves_per_simd;
    k<K, FILL, SHAPE><<<256, threads>>>(out, 10, cyc);
    hipDeviceSynchronize();
    hipMemset(cyc, 0, 8);
    k<K, FILL, SHAPE><<<256, threads>>>(out, iters, cyc);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    return (double)c / (iters * 8.0) / waves_per_simd;  // cycles per group per SIMD (wave 0's view / waves)
}

template <int FILL, int SHAPE>
void sweep(const char* name) {
    for (int w : {1, 2, 4}) {
        printf("%-28s w/SIMD=%d  K=0..8:", name, w);
        printf(" %5.1f", run<0, FILL, SHAPE>(w)); printf(" %5.1f", run<1, FILL, SHAPE>(w));
        printf(" %5.1f", run<2, FILL, SHAPE>(w)); printf(" %5.1f", run<3, FILL, SHAPE>(w));
        printf(" %5.1f", run<4, FILL, SHAPE>(w)); printf(" %5.1f", run<5, FILL, SHAPE>(w));
        printf(" %5.1f", run<6, FILL, SHAPE>(w)); printf(" %5.1f", run<8, FILL, SHAPE>(w));
        printf(" %5.1f", run<12, FILL, SHAPE>(w)); printf(" %5.1f\n", run<16, FILL, SHAPE>(w));
    }
}

int main() {
    printf("cycles per group (1 MFMA + K fillers) per SIMD; K = 0 1 2 3 4 5 6 8 12 16\n");
    sweep<0, 0>("16x16x32 + v_fma_f32");
    sweep<3, 0>("16x16x32 + v_pk_mul_f32");
    sweep<1, 0>("16x16x32 + v_exp_f32");
    sweep<2, 0>("16x16x32 + v_cvt_pk_f16_f32");
    sweep<0, 1>("32x32x16 + v_fma_f32");
    sweep<1, 1>("32x32x16 + v_exp_f32");
    return 0;
}
