// How do the two fp16 MFMA shapes the flow kernels use round their fp32 accumulation?  (round 6: the 16-query kernels,
// v_mfma_f32_16x16x32_f16, read 2-3x the 32-query kernels' error on a few sets with IDENTICAL operand arithmetic, and a signed
// mean of the error that the 32-query kernels, v_mfma_f32_32x32x16_f16, do not show.)
// Every row of A and every column of B is the same vector, so every output element is the same dot product C + sum_k a_k b_k and
// no output layout needs to be known; lane l holds k = 8 (l / ROWS) + j in element j of both fragments (A and B agree, which is
// all a contraction needs).  C = 1536 = 1.5 x 2^10, ulp u = 2^-13; products are exact multiples of u / 4 (or much smaller).
// Prints (result - C) / u per test and shape, next to what round-to-nearest-even of the EXACT sum and truncation toward zero give.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Test {
    const char* name;
    float a[32], b[32];   // k-indexed; the 32x32x16 shape uses k < 16
    float c;
};

__global__ void run16(const _Float16* a, const _Float16* b, float c, float* out) {   // 16x16x32: K = 32, lane l: k = 8 (l / 16) + j
    const int l = threadIdx.x, kb = 8 * (l / 16);
    f16x8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = a[kb + j]; bv[j] = b[kb + j]; }
    f32x4 acc = {c, c, c, c};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
}
__global__ void run32(const _Float16* a, const _Float16* b, float c, float* out) {   // 32x32x16: K = 16, lane l: k = 8 (l / 32) + j
    const int l = threadIdx.x, kb = 8 * (l / 32);
    f16x8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = a[kb + j]; bv[j] = b[kb + j]; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[l * 16 + r] = acc[r];
}
// the same sum through TWO chained MFMAs (accumulator fed back), the second half of k in the second one: how a K = 32 contraction
// runs on the 32x32x16 shape
__global__ void run32x2(const _Float16* a, const _Float16* b, float c, float* out) {
    const int l = threadIdx.x, kb = 8 * (l / 32);
    f16x8 a0, b0, a1, b1;
    for (int j = 0; j < 8; ++j) { a0[j] = a[kb + j]; b0[j] = b[kb + j]; a1[j] = a[16 + kb + j]; b1[j] = b[16 + kb + j]; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[l * 16 + r] = acc[r];
}

static float rz(double x) {   // fp32 truncation toward zero
    float f = (float)x;
    if (std::fabs((double)f) > std::fabs(x)) f = std::nextafterf(f, 0.0f);
    return f;
}

int main() {
    const float C = 1536.0f, U = std::ldexp(1.0f, -13);
    const float q = std::ldexp(1.0f, -14);   // b = 2^-14 (smallest normal fp16): a x q = a x u / 2
    std::vector<Test> tests;
    auto mk = [&](const char* name) { Test t; std::memset(&t, 0, sizeof t); t.name = name; t.c = C; return t; };
    { Test t = mk("one product +0.75 u"); t.a[0] = 1.5f; t.b[0] = q; tests.push_back(t); }
    { Test t = mk("one product -0.25 u"); t.a[0] = -0.5f; t.b[0] = q; tests.push_back(t); }
    { Test t = mk("one product +0.25 u"); t.a[0] = 0.5f; t.b[0] = q; tests.push_back(t); }
    { Test t = mk("one product -0.75 u"); t.a[0] = -1.5f; t.b[0] = q; tests.push_back(t); }
    { Test t = mk("one product +0.5 u (tie, C even)"); t.a[0] = 1.0f; t.b[0] = q; tests.push_back(t); }
    { Test t = mk("+0.5 u and +2^-15 u (above the tie)"); t.a[0] = 1.0f; t.b[0] = q; t.a[1] = q; t.b[1] = q; tests.push_back(t); }
    { Test t = mk("4 x +0.25 u, one k-block"); for (int k = 0; k < 4; ++k) { t.a[k] = 0.5f; t.b[k] = q; } tests.push_back(t); }
    { Test t = mk("4 x +0.25 u, k = 0, 8 (other lane groups)"); for (int k = 0; k < 2; ++k) { t.a[8 * k] = 0.5f; t.b[8 * k] = q; t.a[8 * k + 1] = 0.5f; t.b[8 * k + 1] = q; } tests.push_back(t); }
    { Test t = mk("16 x +0.25 u, k < 16"); for (int k = 0; k < 16; ++k) { t.a[k] = 0.5f; t.b[k] = q; } tests.push_back(t); }
    { Test t = mk("16 x +0.125 u, k < 16"); for (int k = 0; k < 16; ++k) { t.a[k] = 0.25f; t.b[k] = q; } tests.push_back(t); }
    { Test t = mk("+0.75 u, -0.5 u"); t.a[0] = 1.5f; t.b[0] = q; t.a[1] = -1.0f; t.b[1] = q; tests.push_back(t); }
    { Test t = mk("+1024 u, -1023.75 u (cancel to +0.25 u)"); t.a[0] = 2048.0f; t.b[0] = q; t.a[1] = -2047.5f; t.b[1] = q; tests.push_back(t); }
    { Test t = mk("32 x +0.25 u, all k (16x16x32 / two chained 32x32x16)"); for (int k = 0; k < 32; ++k) { t.a[k] = 0.5f; t.b[k] = q; } tests.push_back(t); }
    { Test t = mk("+0.75 u in k = 0 and in k = 16"); t.a[0] = 1.5f; t.b[0] = q; t.a[16] = 1.5f; t.b[16] = q; tests.push_back(t); }
    { Test t = mk("C = 0: 3 x 2^-14 x (1 + 2^-10)^2 (products exact in fp32?)"); t.c = 0.0f; for (int k = 0; k < 3; ++k) { t.a[k] = 1.0f + std::ldexp(1.0f, -10); t.b[k] = q * (1.0f + std::ldexp(1.0f, -10)); } tests.push_back(t); }

    _Float16 *da, *db;
    float* dout;
    hipMalloc(&da, 32 * sizeof(_Float16)); hipMalloc(&db, 32 * sizeof(_Float16)); hipMalloc(&dout, 64 * 16 * sizeof(float));
    std::printf("%-62s %12s %12s | %10s %10s %10s\n", "test: (result - C) / u", "RN(exact)", "RZ(exact)", "16x16x32", "32x32x16", "32x32x16x2");
    for (const Test& t : tests) {
        _Float16 ha[32], hb[32];
        double exact16 = t.c, exact32 = t.c;
        for (int k = 0; k < 32; ++k) {
            ha[k] = (_Float16)t.a[k]; hb[k] = (_Float16)t.b[k];
            exact32 += (double)(float)ha[k] * (double)(float)hb[k];
            if (k < 16) exact16 += (double)(float)ha[k] * (double)(float)hb[k];
        }
        hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
        float h[64 * 16];
        float r16, r32, r32x2;
        run16<<<1, 64>>>(da, db, t.c, dout); hipMemcpy(h, dout, 64 * 4 * sizeof(float), hipMemcpyDeviceToHost); r16 = h[0];
        for (int i = 1; i < 64 * 4; ++i) if (h[i] != r16) std::printf("  (16x16x32: element %d differs: %.9g vs %.9g)\n", i, h[i], r16);
        run32<<<1, 64>>>(da, db, t.c, dout); hipMemcpy(h, dout, 64 * 16 * sizeof(float), hipMemcpyDeviceToHost); r32 = h[0];
        for (int i = 1; i < 64 * 16; ++i) if (h[i] != r32) { std::printf("  (32x32x16: element %d differs: %.9g vs %.9g)\n", i, h[i], r32); break; }
        run32x2<<<1, 64>>>(da, db, t.c, dout); hipMemcpy(h, dout, 64 * 16 * sizeof(float), hipMemcpyDeviceToHost); r32x2 = h[0];
        const double s = t.c != 0.0f ? U : std::ldexp(1.0, -14 - 23);   // the C = 0 test: in units of ulp(2^-14)
        std::printf("%-62s %12.4f %12.4f | %10.4f %10.4f %10.4f   (K = 16 exact: RN %.4f RZ %.4f)\n", t.name, ((double)(float)exact32 - t.c) / s,
                    ((double)rz(exact32) - t.c) / s, ((double)r16 - t.c) / s, ((double)r32 - t.c) / s, ((double)r32x2 - t.c) / s,
                    ((double)(float)exact16 - t.c) / s, ((double)rz(exact16) - t.c) / s);
    }
    return 0;
}
