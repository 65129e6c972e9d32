// Does the fp16 MFMA honour fp16 subnormal inputs on gfx950, and does v_cvt_pk_f16_f32 produce them?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out, float tiny, float big) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.f; b[j] = (_Float16)0.f; }
    a[0] = (_Float16)tiny;   // 2^-20: fp16 subnormal
    b[0] = (_Float16)big;    // 2^10
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
    // B-side subnormal
    f32x4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, d, 0, 0, 0);
    if (threadIdx.x == 0) out[2] = d[0];
}
int main() {
    float* o; hipMalloc(&o, 64);
    k<<<1, 64>>>(o, 9.5367431640625e-07f, 1024.0f);
    float h[3]; hipMemcpy(h, o, 12, hipMemcpyDeviceToHost);
    printf("A-subnormal * 2^10 = %g (expect 0.000976562 if honoured)  cvt(2^-20) = %g  B-subnormal: %g\n", h[0], h[1], h[2]);
    return 0;
}
