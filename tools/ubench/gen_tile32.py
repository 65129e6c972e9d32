#!/usr/bin/env python3
"""Generates tile32.hip: a replay of the shipped disk Euler step's instruction stream (the opcode sequence of
flow_kernel<0,2,2,true,3,false>'s loop, in its real order, as inline asm the compiler cannot re-interleave) in two forms:

  T16: as shipped - a wave owns 16 queries, 38 v_mfma_f32_16x16x32_f16 + 2 v_mfma_f32_16x16x4_f32 + 270 VALU per step;
  T32: what a 32-query tile would issue - every VALU instruction twice (twice the activations per lane), the same NUMBER of
       MFMAs but v_mfma_f32_32x32x16_f16 / v_mfma_f32_32x32x2_f32 (twice the flops each), the same LDS reads and swaps.

Question (round 4, tools/ubench/mfma_src `shapes`): a 32x32x16 MFMA hides ~8 cycles of other waves' VALU work, a 16x16x32 none -
is a 32-query tile worth a rewrite?  Output: cycles per step per SIMD, and per 16 queries, at 1 / 2 / 3 waves per SIMD; plus the
VALU-only and MFMA-only parts of T16.  Register pools rotate, so no instruction depends on a recent one (optimistic for both).
Usage: python gen_tile32.py > tile32.hip"""

OPS = """v_mfma_f32_16x16x4_f32 ds_read_b128 ds_read_b128 ds_read_b128 ds_read_b128 ds_read_b128 s_nop_0 v_exp_f32 v_exp_f32
v_exp_f32 v_exp_f32 v_add_f32 v_add_f32 v_rcp_f32 v_rcp_f32 v_mfma_f32_16x16x4_f32 v_add_f32 v_add_f32 v_rcp_f32 v_rcp_f32
v_mov_b64 v_pk_mul_f32 v_pk_fma_f32 s_nop_0 v_exp_f32 v_pk_fma_f32 v_and_b32 v_and_b32 v_pk_fma_f32 v_pk_mul_f32 v_pk_fma_f32
v_add_f32 v_pk_fma_f32 v_and_b32 v_and_b32 v_pk_fma_f32 v_exp_f32 v_exp_f32 v_exp_f32 v_rcp_f32 v_add_f32 v_add_f32 v_rcp_f32
v_rcp_f32 v_add_f32 v_rcp_f32 v_cvt_pk_f16_f32 v_pk_mul_f32 v_pk_fma_f32 v_cvt_pk_f16_f32 v_pk_fma_f32 v_and_b32 v_and_b32
v_pk_fma_f32 v_pk_mul_f32 v_pk_fma_f32 v_and_b32 v_pk_fma_f32 v_and_b32 v_and_b32 v_cvt_pk_f16_f32 v_pk_fma_f32 ds_read_b128
ds_read_b128 ds_read_b128 ds_read_b128 ds_read_b128 v_and_b32 v_and_b32 v_and_b32 v_and_b32 v_and_b32 v_and_b32 v_and_b32
ds_read_b128 v_cvt_pk_f16_f32 v_cvt_pk_f16_f32 v_cvt_pk_f16_f32 v_cvt_pk_f16_f32 v_cvt_pk_f16_f32 ds_read_b128 s_waitcnt
v_cvt_pk_f16_f32 v_mfma_f32_16x16x32_f16 v_cvt_pk_f16_f32 v_cvt_pk_f16_f32 v_cvt_pk_f16_f32 v_pk_add_f32
v_mfma_f32_16x16x32_f16 v_add_f64 v_mfma_f32_16x16x32_f16 v_mfma_f32_16x16x32_f16 v_add_f32 v_add_f32 v_cvt_pk_f16_f32
v_cvt_pk_f16_f32 v_pk_add_f32 s_nop_0 v_cvt_pk_f16_f32 v_pk_add_f32 s_nop_0 v_cvt_pk_f16_f32 s_waitcnt s_nop_0
v_mfma_f32_16x16x32_f16 v_mfma_f32_16x16x32_f16 v_mfma_f32_16x16x32_f16 v_mfma_f32_16x16x32_f16 v_mfma_f32_16x16x32_f16
v_mfma_f32_16x16x32_f16 v_mfma_f32_16x16x32_f16 s_nop_5 v_exp_f32 v_exp_f32 v_mfma_f32_16x16x32_f16 v_exp_f32 v_exp_f32
v_add_f32 v_mfma_f32_16x16x32_f16 v_add_f32 v_rcp_f32 v_rcp_f32 v_mfma_f32_16x16x32_f16 v_add_f32 v_add_f32 v_rcp_f32
v_mfma_f32_16x16x32_f16 v_rcp_f32 v_exp_f32 v_exp_f32 v_mfma_f32_16x16x32_f16 ds_read_b128 ds_read_b128 v_add_f32
v_mfma_f32_16x16x32_f16 v_rcp_f32 v_mfma_f32_16x16x32_f16 s_nop_7 v_mul_f32 v_mul_f32 v_and_b32 v_and_b32 v_pk_fma_f32 s_nop_0
v_cvt_pk_f16_f32 v_pk_mul_f32 s_nop_0 v_and_b32 v_and_b32 v_pk_fma_f32 v_cvt_pk_f16_f32 v_cvt_pk_f16_f32 v_exp_f32 v_add_f32
v_exp_f32 v_rcp_f32 v_add_f32 v_rcp_f32 v_add_f32 v_rcp_f32 v_cvt_pk_f16_f32 v_pk_mul_f32 v_pk_mul_f32 v_and_b32 v_and_b32
v_pk_fma_f32 v_and_b32 v_and_b32 v_cvt_pk_f16_f32 v_cvt_pk_f16_f32 v_pk_fma_f32 ds_read_b128 v_cvt_pk_f16_f32
v_cvt_pk_f16_f32 ds_read_b128 s_nop_0 s_waitcnt s_nop_0 v_mfma_f32_16x16x32_f16 v_mfma_f32_16x16x32_f16
v_mfma_f32_16x16x32_f16 v_mfma_f32_16x16x32_f16 v_mfma_f32_16x16x32_f16 s_nop_5 v_exp_f32 s_nop_0 v_add_f32 v_rcp_f32 v_exp_f32
v_mfma_f32_16x16x32_f16 v_exp_f32 v_exp_f32 v_add_f32 v_rcp_f32 v_add_f32 v_add_f32 v_rcp_f32 s_nop_0 v_exp_f32 v_exp_f32
v_rcp_f32 v_pk_mul_f32 v_add_f32 v_add_f32 v_pk_mul_f32 v_pk_fma_f32 v_pk_fma_f32 v_rcp_f32 v_rcp_f32 v_pk_fma_f32 v_pk_fma_f32
v_and_b32 v_and_b32 v_and_b32 v_and_b32 v_pk_add_f32 v_pk_add_f32 v_cvt_pk_f16_f32 v_cvt_pk_f16_f32 v_pk_mul_f32 v_pk_fma_f32
v_exp_f32 v_exp_f32 v_pk_fma_f32 v_cvt_pk_f16_f32 v_and_b32 v_and_b32 v_pk_add_f32 v_add_f32 v_cvt_pk_f16_f32 v_add_f32
v_rcp_f32 v_rcp_f32 v_cvt_pk_f16_f32 v_cvt_pk_f16_f32 v_pk_mul_f32 v_pk_fma_f32 s_nop_0 v_pk_fma_f32 s_nop_0 v_and_b32 v_and_b32
v_pk_add_f32 v_cvt_pk_f16_f32 v_cvt_pk_f16_f32 ds_read_b128 ds_read_b128 ds_read_b128 ds_read_b128 ds_read_b128 ds_read_b128
ds_read_b128 ds_read_b128 ds_read_b128 s_nop_0 s_waitcnt s_nop_0 v_mfma_f32_16x16x32_f16 v_mfma_f32_16x16x32_f16
v_mfma_f32_16x16x32_f16 v_mfma_f32_16x16x32_f16 v_mfma_f32_16x16x32_f16 v_mfma_f32_16x16x32_f16 v_mfma_f32_16x16x32_f16
v_mfma_f32_16x16x32_f16 v_mfma_f32_16x16x32_f16 v_mfma_f32_16x16x32_f16 v_mfma_f32_16x16x32_f16 v_mfma_f32_16x16x32_f16
s_nop_5 v_fma_f32 v_fma_f32 v_pk_fma_f32 v_pk_fma_f32 v_pk_mul_f32 v_pk_fma_f32 v_pk_fma_f32 v_pk_fma_f32 v_pk_fma_f32 v_and_b32
v_and_b32 v_and_b32 v_and_b32 v_pk_mul_f32 v_pk_fma_f32 v_pk_fma_f32 v_pk_fma_f32 v_cvt_pk_f16_f32 v_cvt_pk_f16_f32 v_pk_mul_f32
v_pk_fma_f32 v_pk_mul_f32 v_pk_fma_f32 v_pk_fma_f32 v_pk_fma_f32 v_pk_fma_f32 v_pk_fma_f32 v_pk_fma_f32 v_pk_fma_f32 v_pk_mul_f32
v_pk_mul_f32 v_pk_fma_f32 v_pk_fma_f32 v_pk_fma_f32 v_pk_fma_f32 v_pk_mul_f32 v_pk_mul_f32 v_pk_fma_f32 v_pk_fma_f32 v_pk_fma_f32
v_pk_fma_f32 v_pk_add_f32 v_pk_add_f32 v_pk_add_f32 v_pk_add_f32 v_permlane32_swap_b32 s_nop_0 v_permlane32_swap_b32 v_add_f32
v_add_f32 s_nop_0 v_permlane16_swap_b32 v_add_f32 v_fma_f32 v_and_b32 v_and_b32 v_and_b32 v_and_b32 v_mov_b32 v_pk_fma_f32
v_cvt_pk_f16_f32 v_cvt_pk_f16_f32 v_permlane32_swap_b32 s_nop_1 v_mfma_f32_16x16x32_f16 v_mul_f32 v_mov_b32 v_pk_fma_f32 s_nop_0
v_permlane16_swap_b32 v_cvt_pk_f16_f32 v_cvt_pk_f16_f32 v_cvt_pk_f16_f32 v_sub_f32 v_cvt_pk_f16_f32 v_rcp_f32 s_nop_3
v_mfma_f32_16x16x32_f16 v_cndmask_b32 v_mul_f32 s_nop_2 v_pk_add_f32 s_nop_0 v_pk_fma_f32""".split()

NA, NP, NH, NF = 32, 16, 8, 8   # pools: f32 scalars, f32 pairs, packed-f16 words, weight fragments (T16 uses 24 / 12 of a / p)
# ("t16w4": the T16 stream with pools small enough for 128 VGPRs, i.e. 4 waves/SIMD - what a register diet could reach at best)


class Gen:
    def __init__(self, variant):
        self.v = variant           # "t16", "t32", "valu", "mfma"
        self.ia = self.ip = self.ih = self.ifr = self.im = 0
        self.na, self.np_ = (NA, NP) if variant == "t32" else ((20, 10) if variant == "t16w4" else (24, 12))
        self.nf = 6 if variant == "t16w4" else NF
        self.nc = 4 if variant == "t16w4" else 6
        self.out = []

    def a(self):
        self.ia = (self.ia + 1) % self.na
        return self.ia

    def p(self):
        self.ip = (self.ip + 1) % self.np_
        return self.ip

    def emit(self, s):
        self.out.append("            " + s)

    def valu(self, op):
        if op in ("v_exp_f32", "v_rcp_f32"):
            self.emit('asm volatile("%s %%0, %%0" : "+v"(a[%d]));' % (op, self.a()))
        elif op == "v_add_f32":
            self.emit('asm volatile("v_add_f32 %%0, 1.0, %%0" : "+v"(a[%d]));' % self.a())
        elif op in ("v_mul_f32", "v_sub_f32"):
            i = self.a()
            self.emit('asm volatile("%s %%0, %%0, %%1" : "+v"(a[%d]) : "v"(a[%d]));' % (op, i, (i + 7) % self.na))
        elif op == "v_fma_f32":
            i = self.a()
            self.emit('asm volatile("v_fma_f32 %%0, %%0, %%1, %%0" : "+v"(a[%d]) : "v"(a[%d]));' % (i, (i + 7) % self.na))
        elif op == "v_and_b32":
            self.emit('asm volatile("v_and_b32 %%0, 0xffffe000, %%0" : "+v"(a[%d]));' % self.a())
        elif op == "v_mov_b32":
            i = self.a()
            self.emit('asm volatile("v_mov_b32 %%0, %%1" : "=v"(a[%d]) : "v"(a[%d]));' % (i, (i + 7) % self.na))
        elif op == "v_cndmask_b32":
            i = self.a()
            self.emit('asm volatile("v_cndmask_b32 %%0, %%0, %%1, vcc" : "+v"(a[%d]) : "v"(a[%d]));' % (i, (i + 7) % self.na))
        elif op == "v_cvt_pk_f16_f32":
            i = self.a()
            self.ih = (self.ih + 1) % NH
            self.emit('asm volatile("v_cvt_pk_f16_f32 %%0, %%1, %%2" : "=v"(h[%d]) : "v"(a[%d]), "v"(a[%d]));' % (self.ih, i, (i + 5) % self.na))
        elif op in ("v_pk_fma_f32",):
            i = self.p()
            self.emit('asm volatile("v_pk_fma_f32 %%0, %%0, %%1, %%0" : "+v"(p[%d]) : "v"(p[%d]));' % (i, (i + 5) % self.np_))
        elif op in ("v_pk_mul_f32", "v_pk_add_f32"):
            i = self.p()
            self.emit('asm volatile("%s %%0, %%0, %%1" : "+v"(p[%d]) : "v"(p[%d]));' % (op, i, (i + 5) % self.np_))
        elif op == "v_mov_b64":
            i = self.p()
            self.emit('asm volatile("v_mov_b64 %%0, %%1" : "=v"(p[%d]) : "v"(p[%d]));' % (i, (i + 5) % self.np_))
        elif op == "v_add_f64":
            i = self.p()
            self.emit('asm volatile("v_add_f64 %%0, %%0, %%1" : "+v"(p[%d]) : "v"(p[%d]));' % (i, (i + 5) % self.np_))
        elif op.startswith("v_permlane"):
            i = self.a()
            self.emit('asm volatile("%s %%0, %%1" : "+v"(a[%d]), "+v"(a[%d]));' % (op, i, (i + 7) % self.na))
        else:
            raise SystemExit("unhandled " + op)

    def step(self):
        for op in OPS:
            if op.startswith("s_nop_"):
                self.emit('asm volatile("s_nop %s");' % op[6:])
            elif op == "s_waitcnt":
                if self.v != "valu":
                    self.emit('asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");')
            elif op == "ds_read_b128":
                if self.v != "valu":
                    self.ifr = (self.ifr + 1) % self.nf
                    self.emit('asm volatile("ds_read_b128 %%0, %%1 offset:%d" : "=v"(f[%d]) : "v"(lds_base) : "memory");' % (1024 * self.ifr, self.ifr))
            elif op.startswith("v_mfma"):
                if self.v == "valu":
                    continue
                self.im += 1
                fr, hb = self.im % self.nf, self.im % 2
                if op == "v_mfma_f32_16x16x32_f16":
                    if self.v == "t32":
                        self.emit('asm volatile("v_mfma_f32_32x32x16_f16 %%0, %%1, %%2, %%0" : "+v"(C[%d]) : "v"(f[%d]), "v"(hb[%d]));' % ((self.im // 6) % 2, fr, hb))
                    else:
                        self.emit('asm volatile("v_mfma_f32_16x16x32_f16 %%0, %%1, %%2, %%0" : "+v"(c[%d]) : "v"(f[%d]), "v"(hb[%d]));' % ((self.im // 3) % self.nc, fr, hb))
                else:
                    if self.v == "t32":
                        self.emit('asm volatile("v_mfma_f32_32x32x2_f32 %%0, %%1, %%2, %%0" : "+v"(C[%d]) : "v"(w1[0]), "v"(w1[1]));' % (self.im % 2))
                    else:
                        self.emit('asm volatile("v_mfma_f32_16x16x4_f32 %%0, %%1, %%2, %%0" : "+v"(c[%d]) : "v"(w1[0]), "v"(w1[1]));' % (self.im % self.nc))
            else:
                if self.v == "mfma":
                    continue
                self.valu(op)
                if self.v == "t32" and not op.startswith("v_permlane"):
                    self.valu(op)
        return self.out


VARIANTS = ["t16", "t32", "valu", "mfma", "t16w4"]
src = ['// generated by gen_tile32.py - see its docstring', '#include <hip/hip_runtime.h>', '#include <cstdio>',
       'typedef float f32x2 __attribute__((ext_vector_type(2)));', 'typedef float f32x4 __attribute__((ext_vector_type(4)));',
       'typedef float f32x16 __attribute__((ext_vector_type(16)));', 'typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));',
       'template <int V> __global__ __launch_bounds__(V == 4 ? 1024 : 768) void k(float* out, int iters, long long* cyc) {',
       '    __shared__ f16x8 lds[1024];',
       '    for (int j = threadIdx.x; j < 1024; j += blockDim.x) for (int e = 0; e < 8; ++e) lds[j][e] = (_Float16)(0.001f * (j + e));',
       '    __syncthreads();',
       '    const unsigned lds_base = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;',
       '    float a[%d]; for (int j = 0; j < %d; ++j) a[j] = threadIdx.x * 1e-3f + j * 0.01f;' % (NA, NA),
       '    f32x2 p[%d]; for (int j = 0; j < %d; ++j) p[j] = (f32x2){a[j], a[j + 1]};' % (NP, NP),
       '    unsigned h[%d]; for (int j = 0; j < %d; ++j) h[j] = j;' % (NH, NH),
       '    f16x8 f[%d], hb[2]; for (int j = 0; j < %d; ++j) for (int e = 0; e < 8; ++e) { f[j][e] = (_Float16)(a[j] + e); hb[j & 1][e] = (_Float16)(a[e] - j); }' % (NF, NF),
       '    float w1[2] = {a[3], a[4]};',
       '    f32x4 c[6]; if (V != 1) for (int j = 0; j < 6; ++j) c[j] = (f32x4){a[0], a[1], a[2], a[3]};',
       '    f32x16 C[2]; if (V == 1) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) C[j][e] = a[e];',
       '    long long t0 = __builtin_readcyclecounter();', '    for (int i = 0; i < iters; ++i) {']
for vi, v in enumerate(VARIANTS):
    src.append('        if (V == %d) {' % vi)
    src += Gen(v).step()
    src.append('        }')
src += ['    }', '    long long t1 = __builtin_readcyclecounter();',
        '    float s = 0; for (int j = 0; j < %d; ++j) s += a[j]; for (int j = 0; j < %d; ++j) s += p[j][0] + p[j][1];' % (NA, NP),
        '    for (int j = 0; j < %d; ++j) s += (float)h[j];' % NH,
        '    if (V != 1) for (int j = 0; j < 6; ++j) s += c[j][0] + c[j][3];', '    if (V == 1) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += C[j][e];',
        '    out[blockIdx.x * blockDim.x + threadIdx.x] = s;',
        '    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicMax((unsigned long long*)cyc, (unsigned long long)(t1 - t0));', '}',
        'template <int V> void run(const char* name, int queries) {', '    static float* out = nullptr; static long long* cyc = nullptr;',
        '    if (!out) { hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8); }', '    for (int w : {1, 2, 3, 4}) {', '        if (w == 4 && V != 4) continue;',
        '        k<V><<<256, 256 * w>>>(out, 10, cyc); hipDeviceSynchronize(); hipMemset(cyc, 0, 8);',
        '        k<V><<<256, 256 * w>>>(out, 2000, cyc); hipDeviceSynchronize();',
        '        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);',
        '        const double per = (double)c / 2000 / w;',
        '        printf("%-34s waves/SIMD=%d : %7.1f cycles per step per SIMD = %7.1f per 16 queries\\n", name, w, per, per * 16 / queries);',
        '    }', '}',
        'int main() {',
        '    run<0>("T16 (16x16x32, as shipped)", 16); run<1>("T32 (32x32x16, VALU doubled)", 32);',
        '    run<2>("T16 VALU only", 16); run<3>("T16 MFMA + LDS only", 16);',
        '    run<4>("T16, pools for 4 waves/SIMD", 16); return 0;', '}']
print("\n".join(src))
