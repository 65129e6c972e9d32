// VGPR operand-bank conflicts and literal operands: do they change the issue cost of the flow kernel's VALU mix?
// Fixed physical registers through inline asm (sources v1..v16, destinations rotate over v20..v35); 4 waves/SIMD;
// shader cycles per wave64 instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(A) A A A A A A A A A A A A A A A A
#define CLOB "v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35"

template <int P>
__global__ __launch_bounds__(1024) void k(float* out, int iters, long long* cyc) {
    asm volatile("v_mov_b32 v1, 1.0\n v_mov_b32 v2, 1.0\n v_mov_b32 v3, 0.5\n v_mov_b32 v4, 1.0\n v_mov_b32 v5, 0.5\n v_mov_b32 v6, 1.0\n v_mov_b32 v7, 1.0\n v_mov_b32 v8, 0.5\n"
                 "v_mov_b32 v9, 1.0\n v_mov_b32 v10, 1.0\n v_mov_b32 v11, 0.5\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n v_mov_b32 v14, 0.5\n v_mov_b32 v15, 1.0\n v_mov_b32 v16, 1.0\n s_mov_b32 s20, 0xffffe000" ::: CLOB, "s20");
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (P == 0) asm volatile(REP16("v_fma_f32 v20, v1, v2, v3\n v_fma_f32 v21, v5, v6, v7\n v_fma_f32 v22, v9, v10, v11\n v_fma_f32 v23, v13, v14, v15\n") ::: CLOB);          // 3 banks
        if (P == 1) asm volatile(REP16("v_fma_f32 v20, v4, v8, v12\n v_fma_f32 v21, v1, v5, v9\n v_fma_f32 v22, v2, v6, v10\n v_fma_f32 v23, v3, v7, v11\n") ::: CLOB);           // 1 bank
        if (P == 2) asm volatile(REP16("v_fma_f32 v20, v4, v8, v3\n v_fma_f32 v21, v1, v5, v10\n v_fma_f32 v22, v2, v6, v11\n v_fma_f32 v23, v3, v7, v12\n") ::: CLOB);            // 2 of 3 in one bank
        if (P == 3) asm volatile(REP16("v_mul_f32 v20, v1, v2\n v_mul_f32 v21, v5, v6\n v_mul_f32 v22, v9, v10\n v_mul_f32 v23, v13, v14\n") ::: CLOB);                           // VOP2, 2 banks
        if (P == 4) asm volatile(REP16("v_mul_f32 v20, v4, v8\n v_mul_f32 v21, v1, v5\n v_mul_f32 v22, v2, v6\n v_mul_f32 v23, v3, v7\n") ::: CLOB);                              // VOP2, 1 bank
        if (P == 5) asm volatile(REP16("v_pk_fma_f32 v[20:21], v[2:3], v[4:5], v[8:9]\n v_pk_fma_f32 v[22:23], v[6:7], v[12:13], v[16:17]\n v_pk_fma_f32 v[24:25], v[10:11], v[4:5], v[12:13]\n v_pk_fma_f32 v[26:27], v[14:15], v[8:9], v[16:17]\n") ::: CLOB, "v17");   // pairs (2,3),(0,1),(0,1)
        if (P == 6) asm volatile(REP16("v_pk_fma_f32 v[20:21], v[2:3], v[6:7], v[10:11]\n v_pk_fma_f32 v[22:23], v[4:5], v[8:9], v[12:13]\n v_pk_fma_f32 v[24:25], v[2:3], v[6:7], v[14:15]\n v_pk_fma_f32 v[26:27], v[4:5], v[8:9], v[16:17]\n") ::: CLOB, "v17");       // all pairs in the same banks
        if (P == 7) asm volatile(REP16("v_pk_fma_f32 v[20:21], v[2:3], v[4:5], v[6:7]\n v_pk_fma_f32 v[22:23], v[8:9], v[10:11], v[12:13]\n v_pk_fma_f32 v[24:25], v[14:15], v[16:17], v[2:3]\n v_pk_fma_f32 v[26:27], v[4:5], v[6:7], v[8:9]\n") ::: CLOB, "v17");       // mixed
        if (P == 8) asm volatile(REP16("v_and_b32 v20, 0xffffe000, v1\n v_and_b32 v21, 0xffffe000, v2\n v_and_b32 v22, 0xffffe000, v3\n v_and_b32 v23, 0xffffe000, v4\n") ::: CLOB);   // 32-bit literal
        if (P == 9) asm volatile(REP16("v_and_b32 v20, s20, v1\n v_and_b32 v21, s20, v2\n v_and_b32 v22, s20, v3\n v_and_b32 v23, s20, v4\n") ::: CLOB);                          // mask in an SGPR
        if (P == 10) asm volatile(REP16("v_cvt_pk_f16_f32 v20, v1, v2\n v_cvt_pk_f16_f32 v21, v5, v6\n v_cvt_pk_f16_f32 v22, v9, v10\n v_cvt_pk_f16_f32 v23, v13, v14\n") ::: CLOB);
        if (P == 11) asm volatile(REP16("v_cvt_pk_f16_f32 v20, v4, v8\n v_cvt_pk_f16_f32 v21, v1, v5\n v_cvt_pk_f16_f32 v22, v2, v6\n v_cvt_pk_f16_f32 v23, v3, v7\n") ::: CLOB);
        if (P == 12) asm volatile(REP16("v_fma_f32 v20, v1, v2, v20\n v_fma_f32 v21, v5, v6, v21\n v_fma_f32 v22, v9, v10, v22\n v_fma_f32 v23, v13, v14, v23\n") ::: CLOB);      // accumulate in place (dependent every 4th)
        if (P == 13) asm volatile(REP16("v_fma_f32 v1, v1, v2, v3\n v_fma_f32 v5, v5, v6, v7\n v_fma_f32 v9, v9, v10, v11\n v_fma_f32 v13, v13, v14, v15\n") ::: CLOB);          // dst = src0
        if (P == 14) asm volatile(REP16("v_exp_f32 v20, v1\n v_exp_f32 v21, v2\n v_exp_f32 v22, v3\n v_exp_f32 v23, v4\n") ::: CLOB);
        if (P == 15) asm volatile(REP16("v_exp_f32 v20, v1\n v_fma_f32 v24, v5, v6, v7\n v_exp_f32 v22, v3\n v_fma_f32 v25, v9, v10, v11\n") ::: CLOB);                           // trans + plain interleaved (co-issue?)
        if (P == 16) asm volatile(REP16("v_exp_f32 v20, v1\n v_fma_f32 v24, v5, v6, v7\n v_fma_f32 v26, v13, v14, v15\n v_fma_f32 v25, v9, v10, v11\n") ::: CLOB);                  // 1 trans + 3 plain
        if (P == 17) asm volatile("v_exp_f32 v20, v1\n v_add_f32 v20, 1.0, v20\n v_rcp_f32 v20, v20\n v_mul_f32 v28, v20, v1\n v_fma_f32 v9, v28, v20, v20\n v_fma_f32 v28, v9, v28, v20\n v_exp_f32 v21, v2\n v_add_f32 v21, 1.0, v21\n v_rcp_f32 v21, v21\n v_mul_f32 v29, v21, v2\n v_fma_f32 v10, v29, v21, v21\n v_fma_f32 v29, v10, v29, v21\n v_exp_f32 v22, v3\n v_add_f32 v22, 1.0, v22\n v_rcp_f32 v22, v22\n v_mul_f32 v30, v22, v3\n v_fma_f32 v11, v30, v22, v22\n v_fma_f32 v30, v11, v30, v22\n v_exp_f32 v23, v4\n v_add_f32 v23, 1.0, v23\n v_rcp_f32 v23, v23\n v_mul_f32 v31, v23, v4\n v_fma_f32 v12, v31, v23, v23\n v_fma_f32 v31, v12, v31, v23\n v_exp_f32 v24, v5\n v_add_f32 v24, 1.0, v24\n v_rcp_f32 v24, v24\n v_mul_f32 v32, v24, v5\n v_fma_f32 v13, v32, v24, v24\n v_fma_f32 v32, v13, v32, v24\n v_exp_f32 v25, v6\n v_add_f32 v25, 1.0, v25\n v_rcp_f32 v25, v25\n v_mul_f32 v33, v25, v6\n v_fma_f32 v14, v33, v25, v25\n v_fma_f32 v33, v14, v33, v25\n v_exp_f32 v26, v7\n v_add_f32 v26, 1.0, v26\n v_rcp_f32 v26, v26\n v_mul_f32 v34, v26, v7\n v_fma_f32 v15, v34, v26, v26\n v_fma_f32 v34, v15, v34, v26\n v_exp_f32 v27, v8\n v_add_f32 v27, 1.0, v27\n v_rcp_f32 v27, v27\n v_mul_f32 v35, v27, v8\n v_fma_f32 v16, v35, v27, v27\n v_fma_f32 v35, v16, v35, v27\n v_exp_f32 v20, v1\n v_add_f32 v20, 1.0, v20\n v_rcp_f32 v20, v20\n v_mul_f32 v28, v20, v1\n v_fma_f32 v9, v28, v20, v20\n v_fma_f32 v28, v9, v28, v20\n v_exp_f32 v21, v2\n v_add_f32 v21, 1.0, v21\n v_rcp_f32 v21, v21\n v_mul_f32 v29, v21, v2\n v_fma_f32 v10, v29, v21, v21\n v_fma_f32 v29, v10, v29, v21\n v_exp_f32 v22, v3\n v_add_f32 v22, 1.0, v22\n v_rcp_f32 v22, v22\n v_mul_f32 v30, v22, v3\n v_fma_f32 v11, v30, v22, v22\n v_fma_f32 v30, v11, v30, v22\n v_exp_f32 v23, v4\n v_add_f32 v23, 1.0, v23\n v_rcp_f32 v23, v23\n v_mul_f32 v31, v23, v4\n v_fma_f32 v12, v31, v23, v23\n v_fma_f32 v31, v12, v31, v23\n v_exp_f32 v24, v5\n v_add_f32 v24, 1.0, v24\n v_rcp_f32 v24, v24\n v_mul_f32 v32, v24, v5\n v_fma_f32 v13, v32, v24, v24\n v_fma_f32 v32, v13, v32, v24\n v_exp_f32 v25, v6\n v_add_f32 v25, 1.0, v25\n v_rcp_f32 v25, v25\n v_mul_f32 v33, v25, v6\n v_fma_f32 v14, v33, v25, v25\n v_fma_f32 v33, v14, v33, v25\n v_exp_f32 v26, v7\n v_add_f32 v26, 1.0, v26\n v_rcp_f32 v26, v26\n v_mul_f32 v34, v26, v7\n v_fma_f32 v15, v34, v26, v26\n v_fma_f32 v34, v15, v34, v26\n v_exp_f32 v27, v8\n v_add_f32 v27, 1.0, v27\n v_rcp_f32 v27, v27\n v_mul_f32 v35, v27, v8\n v_fma_f32 v16, v35, v27, v27\n v_fma_f32 v35, v16, v35, v27\n " ::: CLOB);   // 48 x 2: sigmoid chains unit by unit (trans isolated between plain ops)
        if (P == 18) asm volatile("v_exp_f32 v20, v1\n v_exp_f32 v21, v2\n v_exp_f32 v22, v3\n v_exp_f32 v23, v4\n v_exp_f32 v24, v5\n v_exp_f32 v25, v6\n v_exp_f32 v26, v7\n v_exp_f32 v27, v8\n v_add_f32 v20, 1.0, v20\n v_add_f32 v21, 1.0, v21\n v_add_f32 v22, 1.0, v22\n v_add_f32 v23, 1.0, v23\n v_add_f32 v24, 1.0, v24\n v_add_f32 v25, 1.0, v25\n v_add_f32 v26, 1.0, v26\n v_add_f32 v27, 1.0, v27\n v_rcp_f32 v20, v20\n v_rcp_f32 v21, v21\n v_rcp_f32 v22, v22\n v_rcp_f32 v23, v23\n v_rcp_f32 v24, v24\n v_rcp_f32 v25, v25\n v_rcp_f32 v26, v26\n v_rcp_f32 v27, v27\n v_mul_f32 v28, v20, v1\n v_mul_f32 v29, v21, v2\n v_mul_f32 v30, v22, v3\n v_mul_f32 v31, v23, v4\n v_mul_f32 v32, v24, v5\n v_mul_f32 v33, v25, v6\n v_mul_f32 v34, v26, v7\n v_mul_f32 v35, v27, v8\n v_fma_f32 v9, v28, v20, v20\n v_fma_f32 v10, v29, v21, v21\n v_fma_f32 v11, v30, v22, v22\n v_fma_f32 v12, v31, v23, v23\n v_fma_f32 v13, v32, v24, v24\n v_fma_f32 v14, v33, v25, v25\n v_fma_f32 v15, v34, v26, v26\n v_fma_f32 v16, v35, v27, v27\n v_fma_f32 v28, v9, v28, v20\n v_fma_f32 v29, v10, v29, v21\n v_fma_f32 v30, v11, v30, v22\n v_fma_f32 v31, v12, v31, v23\n v_fma_f32 v32, v13, v32, v24\n v_fma_f32 v33, v14, v33, v25\n v_fma_f32 v34, v15, v34, v26\n v_fma_f32 v35, v16, v35, v27\n v_exp_f32 v20, v1\n v_exp_f32 v21, v2\n v_exp_f32 v22, v3\n v_exp_f32 v23, v4\n v_exp_f32 v24, v5\n v_exp_f32 v25, v6\n v_exp_f32 v26, v7\n v_exp_f32 v27, v8\n v_add_f32 v20, 1.0, v20\n v_add_f32 v21, 1.0, v21\n v_add_f32 v22, 1.0, v22\n v_add_f32 v23, 1.0, v23\n v_add_f32 v24, 1.0, v24\n v_add_f32 v25, 1.0, v25\n v_add_f32 v26, 1.0, v26\n v_add_f32 v27, 1.0, v27\n v_rcp_f32 v20, v20\n v_rcp_f32 v21, v21\n v_rcp_f32 v22, v22\n v_rcp_f32 v23, v23\n v_rcp_f32 v24, v24\n v_rcp_f32 v25, v25\n v_rcp_f32 v26, v26\n v_rcp_f32 v27, v27\n v_mul_f32 v28, v20, v1\n v_mul_f32 v29, v21, v2\n v_mul_f32 v30, v22, v3\n v_mul_f32 v31, v23, v4\n v_mul_f32 v32, v24, v5\n v_mul_f32 v33, v25, v6\n v_mul_f32 v34, v26, v7\n v_mul_f32 v35, v27, v8\n v_fma_f32 v9, v28, v20, v20\n v_fma_f32 v10, v29, v21, v21\n v_fma_f32 v11, v30, v22, v22\n v_fma_f32 v12, v31, v23, v23\n v_fma_f32 v13, v32, v24, v24\n v_fma_f32 v14, v33, v25, v25\n v_fma_f32 v15, v34, v26, v26\n v_fma_f32 v16, v35, v27, v27\n v_fma_f32 v28, v9, v28, v20\n v_fma_f32 v29, v10, v29, v21\n v_fma_f32 v30, v11, v30, v22\n v_fma_f32 v31, v12, v31, v23\n v_fma_f32 v32, v13, v32, v24\n v_fma_f32 v33, v14, v33, v25\n v_fma_f32 v34, v15, v34, v26\n v_fma_f32 v35, v16, v35, v27\n " ::: CLOB);   // the same 96 instructions, transcendentals batched 8 at a time
        if (P == 19) asm volatile("v_exp_f32 v20, v1\n v_exp_f32 v21, v2\n v_exp_f32 v22, v3\n v_exp_f32 v23, v4\n v_add_f32 v20, 1.0, v20\n v_add_f32 v21, 1.0, v21\n v_add_f32 v22, 1.0, v22\n v_add_f32 v23, 1.0, v23\n v_rcp_f32 v20, v20\n v_rcp_f32 v21, v21\n v_rcp_f32 v22, v22\n v_rcp_f32 v23, v23\n v_mul_f32 v28, v20, v1\n v_fma_f32 v9, v28, v20, v20\n v_fma_f32 v28, v9, v28, v20\n v_mul_f32 v29, v21, v2\n v_fma_f32 v10, v29, v21, v21\n v_fma_f32 v29, v10, v29, v21\n v_mul_f32 v30, v22, v3\n v_fma_f32 v11, v30, v22, v22\n v_fma_f32 v30, v11, v30, v22\n v_mul_f32 v31, v23, v4\n v_fma_f32 v12, v31, v23, v23\n v_fma_f32 v31, v12, v31, v23\n v_exp_f32 v24, v5\n v_exp_f32 v25, v6\n v_exp_f32 v26, v7\n v_exp_f32 v27, v8\n v_add_f32 v24, 1.0, v24\n v_add_f32 v25, 1.0, v25\n v_add_f32 v26, 1.0, v26\n v_add_f32 v27, 1.0, v27\n v_rcp_f32 v24, v24\n v_rcp_f32 v25, v25\n v_rcp_f32 v26, v26\n v_rcp_f32 v27, v27\n v_mul_f32 v32, v24, v5\n v_fma_f32 v13, v32, v24, v24\n v_fma_f32 v32, v13, v32, v24\n v_mul_f32 v33, v25, v6\n v_fma_f32 v14, v33, v25, v25\n v_fma_f32 v33, v14, v33, v25\n v_mul_f32 v34, v26, v7\n v_fma_f32 v15, v34, v26, v26\n v_fma_f32 v34, v15, v34, v26\n v_mul_f32 v35, v27, v8\n v_fma_f32 v16, v35, v27, v27\n v_fma_f32 v35, v16, v35, v27\n v_exp_f32 v20, v1\n v_exp_f32 v21, v2\n v_exp_f32 v22, v3\n v_exp_f32 v23, v4\n v_add_f32 v20, 1.0, v20\n v_add_f32 v21, 1.0, v21\n v_add_f32 v22, 1.0, v22\n v_add_f32 v23, 1.0, v23\n v_rcp_f32 v20, v20\n v_rcp_f32 v21, v21\n v_rcp_f32 v22, v22\n v_rcp_f32 v23, v23\n v_mul_f32 v28, v20, v1\n v_fma_f32 v9, v28, v20, v20\n v_fma_f32 v28, v9, v28, v20\n v_mul_f32 v29, v21, v2\n v_fma_f32 v10, v29, v21, v21\n v_fma_f32 v29, v10, v29, v21\n v_mul_f32 v30, v22, v3\n v_fma_f32 v11, v30, v22, v22\n v_fma_f32 v30, v11, v30, v22\n v_mul_f32 v31, v23, v4\n v_fma_f32 v12, v31, v23, v23\n v_fma_f32 v31, v12, v31, v23\n v_exp_f32 v24, v5\n v_exp_f32 v25, v6\n v_exp_f32 v26, v7\n v_exp_f32 v27, v8\n v_add_f32 v24, 1.0, v24\n v_add_f32 v25, 1.0, v25\n v_add_f32 v26, 1.0, v26\n v_add_f32 v27, 1.0, v27\n v_rcp_f32 v24, v24\n v_rcp_f32 v25, v25\n v_rcp_f32 v26, v26\n v_rcp_f32 v27, v27\n v_mul_f32 v32, v24, v5\n v_fma_f32 v13, v32, v24, v24\n v_fma_f32 v32, v13, v32, v24\n v_mul_f32 v33, v25, v6\n v_fma_f32 v14, v33, v25, v25\n v_fma_f32 v33, v14, v33, v25\n v_mul_f32 v34, v26, v7\n v_fma_f32 v15, v34, v26, v26\n v_fma_f32 v34, v15, v34, v26\n v_mul_f32 v35, v27, v8\n v_fma_f32 v16, v35, v27, v27\n v_fma_f32 v35, v16, v35, v27\n " ::: CLOB);     // batched 4 at a time
    }
    long long t1 = __builtin_readcyclecounter();
    float s;
    asm volatile("v_add_f32 %0, v20, v21\n v_add_f32 %0, %0, v22\n v_add_f32 %0, %0, v23" : "=v"(s) :: CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicMax((unsigned long long*)cyc, (unsigned long long)(t1 - t0));
}

template <int P>
void run(const char* name) {
    static float* out = nullptr; static long long* cyc = nullptr;
    if (!out) { hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8); }
    const int per = P >= 17 ? 96 : 64;
    for (int w : {1, 2, 3, 4}) {
        k<P><<<256, 256 * w>>>(out, 10, cyc);
        hipDeviceSynchronize(); hipMemset(cyc, 0, 8);
        const int iters = 400;
        k<P><<<256, 256 * w>>>(out, iters, cyc);
        hipDeviceSynchronize();
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-58s waves/SIMD=%d : %6.2f cycles per instruction per SIMD\n", name, w, (double)c / iters / per / w);
    }
}
int main() {
    run<0>("v_fma_f32, sources in 3 banks");
    run<1>("v_fma_f32, sources in 1 bank");
    run<2>("v_fma_f32, 2 of 3 sources in one bank");
    run<3>("v_mul_f32 (VOP2), 2 banks");
    run<4>("v_mul_f32 (VOP2), 1 bank");
    run<5>("v_pk_fma_f32, source pairs in banks (2,3),(0,1),(0,1)");
    run<6>("v_pk_fma_f32, all source pairs in the same banks");
    run<7>("v_pk_fma_f32, source pairs in distinct banks where possible");
    run<8>("v_and_b32 with a 32-bit literal");
    run<9>("v_and_b32 with the mask in an SGPR");
    run<10>("v_cvt_pk_f16_f32, 2 banks");
    run<11>("v_cvt_pk_f16_f32, 1 bank");
    run<12>("v_fma_f32 accumulate in place");
    run<13>("v_fma_f32 dst = src0");
    run<14>("v_exp_f32");
    run<15>("v_exp_f32 + v_fma_f32 alternating (2 per pair)");
    run<16>("v_exp_f32 + 3 v_fma_f32 (per 4)");
    run<17>("sigmoid chains, unit by unit (16 trans among 96)");
    run<18>("same instructions, trans batched 8 at a time");
    run<19>("same instructions, trans batched 4 at a time");
    return 0;
}
