// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU / MFMA
// instructions the flow kernel is built from, on gfx950.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int WHICH>
__global__ __launch_bounds__(1024) void k(float* out, int iters, long long* cyc) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f32x4 c0 = {a0, a1, a2, a3}, c1 = c0, c2 = c0, c3 = c0;
    f16x8 h0, h1;
    for (int j = 0; j < 8; ++j) { h0[j] = (_Float16)(a0 + j); h1[j] = (_Float16)(a1 - j); }
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (WHICH == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (WHICH == 1) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3" : "+v"(*(double*)&c0), "+v"(*((double*)&c0 + 1)), "+v"(*(double*)&c1), "+v"(*((double*)&c1 + 1)));) }
        if (WHICH == 2) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (WHICH == 3) { REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (WHICH == 4) { REP8(asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n v_cvt_pk_f16_f32 %2, %2, %3\n v_cvt_pk_f16_f32 %4, %4, %5\n v_cvt_pk_f16_f32 %6, %6, %7\n v_cvt_pk_f16_f32 %1, %1, %0\n v_cvt_pk_f16_f32 %3, %3, %2\n v_cvt_pk_f16_f32 %5, %5, %4\n v_cvt_pk_f16_f32 %7, %7, %6" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (WHICH == 5) { REP8(asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%0 op_sel_hi:[0,0,1]\n v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]\n v_fma_mixlo_f16 %2, %3, 1.0, -%2 op_sel_hi:[0,0,1]\n v_fma_mixhi_f16 %2, %3, 1.0, -%0 op_sel_hi:[0,0,1]\n v_fma_mixlo_f16 %4, %5, 1.0, -%4 op_sel_hi:[0,0,1]\n v_fma_mixhi_f16 %4, %5, 1.0, -%6 op_sel_hi:[0,0,1]\n v_fma_mixlo_f16 %6, %7, 1.0, -%6 op_sel_hi:[0,0,1]\n v_fma_mixhi_f16 %6, %7, 1.0, -%4 op_sel_hi:[0,0,1]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (WHICH == 6) { REP8(c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c1, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c3, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c1, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c3, 0, 0, 0);) }
        if (WHICH == 7) {  // 1 MFMA + 4 VALU fma interleaved
            REP8(c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c0, 0, 0, 0); asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
                 c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c1, 0, 0, 0); asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
                 c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c2, 0, 0, 0); asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
                 c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c3, 0, 0, 0); asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
                 c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c0, 0, 0, 0); asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
                 c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c1, 0, 0, 0); asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
                 c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c2, 0, 0, 0); asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
                 c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c3, 0, 0, 0); asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        }
        if (WHICH == 8) { REP8(asm volatile("v_mul_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n v_mul_f32 %4, %4, %4\n v_add_f32 %5, %5, %5\n v_mul_f32 %6, %6, %6\n v_add_f32 %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (WHICH == 9) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3\n v_pk_mul_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3" : "+v"(*(double*)&c0), "+v"(*((double*)&c0 + 1)), "+v"(*(double*)&c1), "+v"(*((double*)&c1 + 1)));) }
        if (WHICH == 10) { REP8(asm volatile("v_cvt_f32_f16 %0, %0\n v_cvt_f32_f16 %1, %1\n v_cvt_f32_f16 %2, %2\n v_cvt_f32_f16 %3, %3\n v_cvt_f32_f16 %4, %4\n v_cvt_f32_f16 %5, %5\n v_cvt_f32_f16 %6, %6\n v_cvt_f32_f16 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (WHICH == 12) { REP8(asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1\n v_cvt_pkrtz_f16_f32 %2, %2, %3\n v_cvt_pkrtz_f16_f32 %4, %4, %5\n v_cvt_pkrtz_f16_f32 %6, %6, %7\n v_cvt_pkrtz_f16_f32 %1, %1, %0\n v_cvt_pkrtz_f16_f32 %3, %3, %2\n v_cvt_pkrtz_f16_f32 %5, %5, %4\n v_cvt_pkrtz_f16_f32 %7, %7, %6" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (WHICH == 13) { REP8(asm volatile("v_and_b32 %0, %0, %1\n v_and_b32 %2, %2, %3\n v_and_b32 %4, %4, %5\n v_and_b32 %6, %6, %7\n v_and_b32 %1, %1, %0\n v_and_b32 %3, %3, %2\n v_and_b32 %5, %5, %4\n v_and_b32 %7, %7, %6" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (WHICH == 14) { REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %1, %1, %0\n v_cvt_pk_bf16_f32 %3, %3, %2\n v_cvt_pk_bf16_f32 %5, %5, %4\n v_cvt_pk_bf16_f32 %7, %7, %6" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (WHICH == 15) { REP8(asm volatile("v_perm_b32 %0, %0, %1, %0\n v_perm_b32 %2, %2, %3, %2\n v_perm_b32 %4, %4, %5, %4\n v_perm_b32 %6, %6, %7, %6\n v_perm_b32 %1, %1, %0, %1\n v_perm_b32 %3, %3, %2, %3\n v_perm_b32 %5, %5, %4, %5\n v_perm_b32 %7, %7, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (WHICH == 11) {  // 1 MFMA + 2 exp interleaved
            REP8(c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c0, 0, 0, 0); asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(a0), "+v"(a1));
                 c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c1, 0, 0, 0); asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(a2), "+v"(a3));
                 c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c2, 0, 0, 0); asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(a4), "+v"(a5));
                 c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c3, 0, 0, 0); asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(a6), "+v"(a7));
                 c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c0, 0, 0, 0); asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(a0), "+v"(a1));
                 c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c1, 0, 0, 0); asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(a2), "+v"(a3));
                 c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c2, 0, 0, 0); asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(a4), "+v"(a5));
                 c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c3, 0, 0, 0); asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(a6), "+v"(a7));)
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0[0] + c1[1] + c2[2] + c3[3] + c0[1] + c0[2] + c0[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int W>
void run(const char* name, int insts_per_iter, int waves_per_simd) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    const int threads = 64 * 4 * waves_per_simd;  // waves_per_simd waves on each of the CU's 4 SIMDs
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<W><<<256, threads>>>(out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<W><<<256, threads>>>(out, iters, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double n = (double)iters * insts_per_iter;           // instructions per wave
    // s_memtime-style counter runs at a fixed 100 MHz on this part; use wall time and assume clock from ms
    double cyc_per_inst_wall = ms * 1e-3 * 2.4e9 / (n * waves_per_simd);
    printf("%-34s waves/SIMD=%d  %8.3f ms  %.2f cyc/inst/SIMD @2.4GHz (counter: %.2f ticks/inst/wave)\n", name, waves_per_simd, ms, cyc_per_inst_wall, (double)c / n);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", 64, w);
        run<8>("v_mul/add_f32", 64, w);
        run<1>("v_pk_fma_f32", 64, w);
        run<9>("v_pk_mul/add_f32", 64, w);
        run<2>("v_exp_f32", 64, w);
        run<3>("v_rcp_f32", 64, w);
        run<4>("v_cvt_pk_f16_f32", 64, w);
        run<10>("v_cvt_f32_f16", 64, w);
        run<5>("v_fma_mixlo/hi_f16", 64, w);
        run<12>("v_cvt_pkrtz_f16_f32", 64, w);
        run<13>("v_and_b32", 64, w);
        run<14>("v_cvt_pk_bf16_f32", 64, w);
        run<15>("v_perm_b32", 64, w);
        run<6>("mfma_16x16x32_f16", 64, w);
        run<7>("mfma + 4 v_fma (per 5 inst)", 64 * 5, w);
        run<11>("mfma + 2 v_exp (per 3 inst)", 64 * 3, w);
    }
    return 0;
}
