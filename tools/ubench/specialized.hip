// Do an MFMA-only wave and a VALU-only wave on the SAME SIMD overlap?  (wave specialisation test)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// mode bit0: mfma waves active, bit1: valu waves active.  layout: waves [0, nm) MFMA, [nm, nm+nv) VALU
template <int filler>
__global__ __launch_bounds__(1024) void k(float* out, int iters, int mode, int nm, long long* cyc, int mfma_mult) {
    const int wave = threadIdx.x >> 6;
    float a[8];
    for (int j = 0; j < 8; ++j) a[j] = threadIdx.x * 1e-3f + j;
    f32x4 c[4];
    for (int j = 0; j < 4; ++j) c[j] = (f32x4){a[0], a[1], a[2], a[3]};
    f16x8 h0, h1;
    for (int j = 0; j < 8; ++j) { h0[j] = (_Float16)(a[0] + j); h1[j] = (_Float16)(a[1] - j); }
    long long t0 = __builtin_readcyclecounter();
    if (wave < nm) {
        if (mode & 1)
            for (int i = 0; i < iters * mfma_mult; ++i) {
#pragma unroll
                for (int u = 0; u < 16; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c[u & 3], 0, 0, 0);
            }
    } else {
        if (mode & 2)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (filler == 0) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
                    else asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
                }
            }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 8; ++j) s += a[j];
    for (int j = 0; j < 4; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicMax((unsigned long long*)cyc, (unsigned long long)(t1 - t0));
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    const int iters = 500;
    for (int filler = 0; filler < 2; ++filler)
      for (int mult : {2, 4, 8})
        for (int nv : {4, 8, 12}) {          // 1,2,3 VALU waves per SIMD next to 1 MFMA wave per SIMD
            const int nm = 4, threads = 64 * (nm + nv);
            long long r[4] = {0, 0, 0, 0};
            for (int mode = 1; mode <= 3; ++mode) {
                if (filler) k<1><<<256, threads>>>(out, 10, mode, nm, cyc, mult); else k<0><<<256, threads>>>(out, 10, mode, nm, cyc, mult);
                hipDeviceSynchronize(); hipMemset(cyc, 0, 8);
                if (filler) k<1><<<256, threads>>>(out, iters, mode, nm, cyc, mult); else k<0><<<256, threads>>>(out, iters, mode, nm, cyc, mult);
                hipDeviceSynchronize(); hipMemcpy(&r[mode], cyc, 8, hipMemcpyDeviceToHost);
            }
            printf("%s x%d: 1 MFMA wave + %d VALU waves per SIMD: mfma-only %lld  valu-only %lld  both %lld cycles  (sum %lld, max %lld)\n",
                   filler ? "v_exp" : "v_fma", mult, nv / 4, r[1], r[2], r[3], r[1] + r[2], r[1] > r[2] ? r[1] : r[2]);
        }
    return 0;
}
