// Micro-benchmark, round 2: issue cost (shader cycles per wave64 instruction per SIMD, 4 waves/SIMD unless said) of
// the instructions considered for the fp16 activation path, the VALU output layer and the cross-lane reductions.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rates2.hip -o valu_rates2
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
// 8 instructions on 8 independent registers; OP uses %0 (dst/src), %1 (second source, another live register)
#define BLOCK8(OP)                                                                                                     \
    asm volatile(OP(0, 1) OP(1, 2) OP(2, 3) OP(3, 4) OP(4, 5) OP(5, 6) OP(6, 7) OP(7, 0)                               \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));

#define OP_EXP16(d, s) "v_exp_f16 %" #d ", %" #d "\n"
#define OP_RCP16(d, s) "v_rcp_f16 %" #d ", %" #d "\n"
#define OP_EXP16_HI(d, s) "v_exp_f16_sdwa %" #d ", %" #d " dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n"
#define OP_RCP16_HI(d, s) "v_rcp_f16_sdwa %" #d ", %" #d " dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n"
#define OP_PKFMA16(d, s) "v_pk_fma_f16 %" #d ", %" #d ", %" #s ", %" #d "\n"
#define OP_PKMUL16(d, s) "v_pk_mul_f16 %" #d ", %" #d ", %" #s "\n"
#define OP_PKADD16(d, s) "v_pk_add_f16 %" #d ", %" #d ", %" #s "\n"
#define OP_SIN(d, s) "v_sin_f32 %" #d ", %" #d "\n"
#define OP_CNDMASK(d, s) "v_cndmask_b32 %" #d ", %" #d ", %" #s ", vcc\n"
#define OP_DOT2(d, s) "v_dot2c_f32_f16 %" #d ", %" #s ", %" #s "\n"
#define OP_PL32(d, s) "v_permlane32_swap_b32 %" #d ", %" #s "\n"
#define OP_PL16(d, s) "v_permlane16_swap_b32 %" #d ", %" #s "\n"
#define OP_DPP_ADD(d, s) "v_add_f32_dpp %" #d ", %" #s ", %" #d " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_FMAC(d, s) "v_fmac_f32 %" #d ", %" #s ", %" #s "\n"
#define OP_FMA(d, s) "v_fma_f32 %" #d ", %" #d ", %" #s ", %" #d "\n"
#define OP_CVT16(d, s) "v_cvt_f16_f32 %" #d ", %" #d "\n"
#define OP_XOR(d, s) "v_xor_b32 %" #d ", %" #d ", %" #s "\n"
#define OP_BFI(d, s) "v_bfi_b32 %" #d ", %" #s ", %" #d ", %" #s "\n"
#define OP_RNDNE(d, s) "v_rndne_f32 %" #d ", %" #d "\n"
#define OP_CVTI(d, s) "v_cvt_i32_f32 %" #d ", %" #d "\n"
#define OP_MED3(d, s) "v_med3_f32 %" #d ", %" #d ", %" #s ", %" #s "\n"
#define OP_LSHLOR(d, s) "v_lshl_or_b32 %" #d ", %" #d ", 16, %" #s "\n"
#define OP_PKMUL32(d, s) "v_pk_mul_f32 %" #d ", %" #d ", %" #s "\n"

template <int W>
__global__ __launch_bounds__(1024) void k(float* out, int iters, long long* cyc) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (W == 0) { REP8(BLOCK8(OP_EXP16)) }
        if (W == 1) { REP8(BLOCK8(OP_RCP16)) }
        if (W == 2) { REP8(BLOCK8(OP_EXP16_HI)) }
        if (W == 3) { REP8(BLOCK8(OP_RCP16_HI)) }
        if (W == 4) { REP8(BLOCK8(OP_PKFMA16)) }
        if (W == 5) { REP8(BLOCK8(OP_PKMUL16)) }
        if (W == 6) { REP8(BLOCK8(OP_PKADD16)) }
        if (W == 7) { REP8(BLOCK8(OP_SIN)) }
        if (W == 8) { REP8(BLOCK8(OP_CNDMASK)) }
        if (W == 9) { REP8(BLOCK8(OP_DOT2)) }
        if (W == 10) { REP8(BLOCK8(OP_PL32)) }
        if (W == 11) { REP8(BLOCK8(OP_PL16)) }
        if (W == 12) { REP8(BLOCK8(OP_DPP_ADD)) }
        if (W == 13) { REP8(BLOCK8(OP_FMAC)) }
        if (W == 14) { REP8(BLOCK8(OP_FMA)) }
        if (W == 15) { REP8(BLOCK8(OP_CVT16)) }
        if (W == 16) { REP8(BLOCK8(OP_XOR)) }
        if (W == 17) { REP8(BLOCK8(OP_BFI)) }
        if (W == 18) { REP8(BLOCK8(OP_RNDNE)) }
        if (W == 19) { REP8(BLOCK8(OP_CVTI)) }
        if (W == 20) { REP8(BLOCK8(OP_MED3)) }
        if (W == 21) { REP8(BLOCK8(OP_LSHLOR)) }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicMax((unsigned long long*)cyc, (unsigned long long)(t1 - t0));
}

template <int W>
void run(const char* name) {
    static float* out = nullptr;
    static long long* cyc = nullptr;
    if (!out) { hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8); }
    for (int w : {1, 4}) {
        const int iters = 1000;
        k<W><<<256, 256 * w>>>(out, 10, cyc);
        hipDeviceSynchronize();
        hipMemset(cyc, 0, 8);
        k<W><<<256, 256 * w>>>(out, iters, cyc);
        hipDeviceSynchronize();
        long long c;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-28s waves/SIMD=%d  %6.2f cycles/inst/SIMD\n", name, w, (double)c / (iters * 64.0) / w);
    }
}

int main() {
    run<14>("v_fma_f32 (reference)");
    run<0>("v_exp_f16");
    run<1>("v_rcp_f16");
    run<2>("v_exp_f16 sdwa hi-half");
    run<3>("v_rcp_f16 sdwa hi-half");
    run<4>("v_pk_fma_f16");
    run<5>("v_pk_mul_f16");
    run<6>("v_pk_add_f16");
    run<7>("v_sin_f32");
    run<8>("v_cndmask_b32");
    run<9>("v_dot2c_f32_f16");
    run<10>("v_permlane32_swap");
    run<11>("v_permlane16_swap");
    run<12>("v_add_f32 dpp row_shr:1");
    run<13>("v_fmac_f32");
    run<15>("v_cvt_f16_f32");
    run<16>("v_xor_b32");
    run<17>("v_bfi_b32");
    run<18>("v_rndne_f32");
    run<19>("v_cvt_i32_f32");
    run<20>("v_med3_f32");
    run<21>("v_lshl_or_b32");
    return 0;
}
