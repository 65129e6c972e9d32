// Micro-benchmark (round 4): what a v_mfma_f32_16x16x32_f16 costs the SIMD's shared issue path by WHERE its operands live.
//
// Round 3 found that an MFMA holds the VALU issue path ~9.6 of its 16 matrix-pipe cycles (tools/ubench/spec2).  If that hold
// is the operand fetch from the VGPR file (A 4 + B 4 + C 4 registers), it should shrink when C is the inline constant 0 or
// when operands come from the AGPR half of the unified file.  Measured here: cycles per group [1 MFMA + K independent plain
// VALU] per SIMD, 3 waves/SIMD (the flow kernel's occupancy), for the operand placements
//   vvv : A, B, C/D in VGPRs          vv0 : C = 0 (D in a VGPR)        avv : A in AGPRs
//   aav : A and B in AGPRs            vva : C/D in AGPRs               aaa : everything in AGPRs
// and, second table, the ORDER of the three split products of two accumulators (what hipcc emits vs what the source asks):
//   dep : c0 c0 c0 c1 c1 c1 (each MFMA reads the accumulator the previous one wrote)      alt : c0 c1 c0 c1 c0 c1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define FMA(x) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x))

template <int MODE, int K>
__global__ __launch_bounds__(768) void k(float* out, int iters, long long* cyc) {
    float a[8];
    for (int j = 0; j < 8; ++j) a[j] = threadIdx.x * 1e-3f + j;
    f32x4 c[4];
    for (int j = 0; j < 4; ++j) c[j] = (f32x4){a[0], a[1], a[2], a[3]};
    f16x8 h0, h1;
    for (int j = 0; j < 8; ++j) { h0[j] = (_Float16)(a[0] + j); h1[j] = (_Float16)(a[1] - j); }
    f32x4 ca[4];  // accumulators in AGPRs (modes 4, 5)
    f32x4 ha0, ha1;  // (128-bit A / B fragments held in AGPRs; the element type is irrelevant to the asm operand)
    for (int j = 0; j < 4; ++j) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ca[j][0]) : "v"(a[j]));
    for (int j = 0; j < 4; ++j) { ca[j][1] = ca[j][0]; ca[j][2] = ca[j][0]; ca[j][3] = ca[j][0]; }
    asm volatile("" : "+a"(ca[0]), "+a"(ca[1]), "+a"(ca[2]), "+a"(ca[3]));
    for (int j = 0; j < 4; ++j) {
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ha0[j]) : "v"(a[j]));
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ha1[j]) : "v"(a[j + 4]));
    }
    asm volatile("" : "+a"(ha0), "+a"(ha1));
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            f32x4& cc = c[u & 3];
            f32x4& cq = ca[u & 3];
            if (MODE == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(cc) : "v"(h0), "v"(h1));
            if (MODE == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(cc) : "v"(h0), "v"(h1));
            if (MODE == 2) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(cc) : "a"(ha0), "v"(h1));
            if (MODE == 3) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(cc) : "a"(ha0), "a"(ha1));
            if (MODE == 4) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(cq) : "v"(h0), "v"(h1));
            if (MODE == 5) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(cq) : "a"(ha0), "a"(ha1));
            if (MODE == 6) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(cc) : "a"(ha0), "v"(h1));
#pragma unroll
            for (int f = 0; f < K; ++f) FMA(a[(u * K + f) & 7]);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 8; ++j) s += a[j];
    for (int j = 0; j < 4; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    for (int j = 0; j < 4; ++j) {
        float t;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(ca[j][1]));
        s += t;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicMax((unsigned long long*)cyc, (unsigned long long)(t1 - t0));
}

// order of the split products: ORDER 0 = dep (c0 c0 c0 c1 c1 c1), 1 = alt (c0 c1 c0 c1 c0 c1); then K plain VALU
template <int ORDER, int K>
__global__ __launch_bounds__(768) void kord(float* out, int iters, long long* cyc) {
    float a[8];
    for (int j = 0; j < 8; ++j) a[j] = threadIdx.x * 1e-3f + j;
    f32x4 c0 = {a[0], a[1], a[2], a[3]}, c1 = c0;
    f16x8 h0, h1, h2;
    for (int j = 0; j < 8; ++j) { h0[j] = (_Float16)(a[0] + j); h1[j] = (_Float16)(a[1] - j); h2[j] = (_Float16)(a[2] * j); }
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (ORDER == 0)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, 0\n v_mfma_f32_16x16x32_f16 %0, %2, %4, %0\n v_mfma_f32_16x16x32_f16 %0, %4, %3, %0\n"
                             "v_mfma_f32_16x16x32_f16 %1, %3, %2, 0\n v_mfma_f32_16x16x32_f16 %1, %3, %4, %1\n v_mfma_f32_16x16x32_f16 %1, %4, %2, %1\n"
                             : "+v"(c0), "+v"(c1) : "v"(h0), "v"(h1), "v"(h2));
            else
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, 0\n v_mfma_f32_16x16x32_f16 %1, %3, %2, 0\n v_mfma_f32_16x16x32_f16 %0, %2, %4, %0\n"
                             "v_mfma_f32_16x16x32_f16 %1, %3, %4, %1\n v_mfma_f32_16x16x32_f16 %0, %4, %3, %0\n v_mfma_f32_16x16x32_f16 %1, %4, %2, %1\n"
                             : "+v"(c0), "+v"(c1) : "v"(h0), "v"(h1), "v"(h2));
#pragma unroll
            for (int f = 0; f < K; ++f) FMA(a[f & 7]);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 8; ++j) s += a[j];
    s += c0[0] + c0[3] + c1[1] + c1[2];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicMax((unsigned long long*)cyc, (unsigned long long)(t1 - t0));
}


// Round 4, second question: does the issue-path hold scale with the MFMA's passes or is it per instruction?  The same group
// test with v_mfma_f32_32x32x16_f16 (twice the flops, 16 passes) and, for reference, v_mfma_f32_16x16x16_f16 (half, 4 passes... on
// gfx950 8) — cycles per [1 MFMA + K v_fma_f32] group per SIMD.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <int SHAPE, int K>
__global__ __launch_bounds__(768) void kshape(float* out, int iters, long long* cyc) {
    float a[8];
    for (int j = 0; j < 8; ++j) a[j] = threadIdx.x * 1e-3f + j;
    f32x16 c[2];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) c[j][i] = a[i & 7];
    f32x4 d[4];
    for (int j = 0; j < 4; ++j) d[j] = (f32x4){a[0], a[1], a[2], a[3]};
    f16x8 h0, h1;
    for (int j = 0; j < 8; ++j) { h0[j] = (_Float16)(a[0] + j); h1[j] = (_Float16)(a[1] - j); }
    f16x4 g0 = {h0[0], h0[1], h0[2], h0[3]}, g1 = {h1[0], h1[1], h1[2], h1[3]};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (SHAPE == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[u & 1]) : "v"(h0), "v"(h1));
            if (SHAPE == 1) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(d[u & 3]) : "v"(g0), "v"(g1));
            if (SHAPE == 2) asm volatile("v_mfma_f32_32x32x8_f16 %0, %1, %2, %0" : "+v"(c[u & 1]) : "v"(g0), "v"(g1));
            if (SHAPE == 3) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d[u & 3]) : "v"(h0), "v"(h1));
#pragma unroll
            for (int f = 0; f < K; ++f) FMA(a[(u * K + f) & 7]);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 8; ++j) s += a[j];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) s += c[j][i];
    for (int j = 0; j < 4; ++j) s += d[j][0] + d[j][1] + d[j][2] + d[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicMax((unsigned long long*)cyc, (unsigned long long)(t1 - t0));
}

static float* g_out = nullptr;
static long long* g_cyc = nullptr;
template <typename F>
double timeit(F launch, int iters, double groups_per_iter, int waves_per_simd) {
    if (!g_out) { hipMalloc(&g_out, 256 * 1024 * 4); hipMalloc(&g_cyc, 8); }
    launch(10);
    hipDeviceSynchronize();
    hipMemset(g_cyc, 0, 8);
    launch(iters);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
    return (double)c / (iters * groups_per_iter) / waves_per_simd;
}

template <int MODE, int K>
double run(int w) { return timeit([&](int it) { k<MODE, K><<<256, 256 * w>>>(g_out, it, g_cyc); }, 1000, 8.0, w); }
template <int ORDER, int K>
double runo(int w) { return timeit([&](int it) { kord<ORDER, K><<<256, 256 * w>>>(g_out, it, g_cyc); }, 1000, 4.0, w); }

template <int SHAPE, int K>
double runs(int w) { return timeit([&](int it) { kshape<SHAPE, K><<<256, 256 * w>>>(g_out, it, g_cyc); }, 1000, 8.0, w); }
template <int SHAPE>
void rowshape(const char* name) {
    printf("%-10s |", name);
    for (int w : {1, 2, 3})
        printf(" w=%d: K=0 %5.1f K=4 %5.1f K=8 %5.1f K=12 %5.1f K=16 %5.1f K=24 %5.1f |", w, runs<SHAPE, 0>(w), runs<SHAPE, 4>(w), runs<SHAPE, 8>(w),
               runs<SHAPE, 12>(w), runs<SHAPE, 16>(w), runs<SHAPE, 24>(w));
    printf("\n");
}

template <int MODE>
void row(const char* name) {
    printf("%-4s |", name);
    for (int w : {1, 3}) printf(" w=%d: K=0 %5.1f  K=2 %5.1f  K=4 %5.1f  K=8 %5.1f |", w, run<MODE, 0>(w), run<MODE, 2>(w), run<MODE, 4>(w), run<MODE, 8>(w));
    printf("\n");
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "pmc0")) {
        // the same VALU stream WITHOUT the MFMAs (what SQ_ACTIVE_INST_VALU reads on pure v_fma_f32 at full rate)
        if (!g_out) { hipMalloc(&g_out, 256 * 1024 * 4); hipMalloc(&g_cyc, 8); }
        hipMemset(g_cyc, 0, 8);
        const int iters = 40000;
        k<7, 8><<<256, 768>>>(g_out, iters, g_cyc);
        hipDeviceSynchronize();
        long long c; hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
        printf("pmc0 run: [8 v_fma_f32] x %d groups per wave, 3 waves/SIMD: %.2f cycles per v_fma_f32 per SIMD\n", iters * 8, (double)c / (iters * 64.0) / 3);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "pmc")) {
        // one saturated configuration, long enough for a counter pass (tools/r04_pmc_ubench.sh): [1 MFMA + 8 v_fma_f32] per group,
        // 3 waves/SIMD, every CU — by construction the SIMD has issuable work in every cycle
        if (!g_out) { hipMalloc(&g_out, 256 * 1024 * 4); hipMalloc(&g_cyc, 8); }
        hipMemset(g_cyc, 0, 8);
        const int iters = 20000;
        k<0, 8><<<256, 768>>>(g_out, iters, g_cyc);
        hipDeviceSynchronize();
        long long c; hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
        printf("pmc run: [1 MFMA + 8 v_fma_f32] x %d groups per wave, 3 waves/SIMD: %.1f cycles per group per SIMD\n", iters * 8, (double)c / (iters * 8.0) / 3);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "shapes")) {
        printf("cycles per [1 MFMA + K v_fma_f32] group per SIMD by MFMA shape (w = waves/SIMD)\n");
        rowshape<3>("16x16x32"); rowshape<0>("32x32x16"); rowshape<1>("16x16x16"); rowshape<2>("32x32x8");
        return 0;
    }
    printf("cycles per [1 MFMA 16x16x32 f16 + K v_fma_f32] group per SIMD (w = waves/SIMD)\n");
    row<0>("vvv"); row<1>("vv0"); row<2>("avv"); row<3>("aav"); row<4>("vva"); row<5>("aaa"); row<6>("av0");
    printf("cycles per [6 MFMAs of two accumulators + K v_fma_f32] group per SIMD\n");
    for (int w : {1, 2, 3}) {
        printf("w=%d dep: K=0 %6.1f K=12 %6.1f K=24 %6.1f | alt: K=0 %6.1f K=12 %6.1f K=24 %6.1f\n", w, runo<0, 0>(w), runo<0, 12>(w), runo<0, 24>(w),
               runo<1, 0>(w), runo<1, 12>(w), runo<1, 24>(w));
    }
    return 0;
}
