// flow_dev.h — device helpers, kernel-parameter block and constants shared by the flow-kernel translation units
// (bsdfd.hip: 16-query tiles on the 16x16 MFMA shapes + the host side; flow32.hip: 32-query tiles on v_mfma_f32_32x32x16_f16).
// Everything lives in an unnamed namespace: each translation unit gets its own copy, nothing is exported.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "bsdfd.h"
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int PE_BANDS = 5;       // rendering/brdf_measured_disk.py:43 (POSITIONAL_ENCODING_BASIS_NUM=5)
constexpr int BASE_PE_BANDS = 3;  // :49
constexpr int BASE_HIDDEN = 16;
constexpr int PE_SLABS = PE_BANDS + 1;  // K=4 slabs: one per band (sin/cos x 2 dims) + the raw (y0,y1) slab
// Split-fp16 operands: x = hi + lo with hi = x truncated to fp16's 11 significant bits (one v_and)
// and lo = fp16(x - hi).  lo may be an fp16 subnormal; the gfx950 MFMA honours fp16 subnormal
// inputs (measured: tools/ubench/denorm.hip), so no rescaling is needed and all three products
// hi*hi + hi*lo + lo*hi accumulate into ONE fp32 accumulator.

enum { OP_SAMPLE = 0, OP_PDF = 1, OP_SAMPLES_ONLY = 2, OP_SAMPLE_PDF = 3 };  // 3: plugin io only, sample(wi) then pdf(wi, wl)
// Precision of the TANGENT contractions: 3 = hi+lo operands and the W_lo product, like the activations.  Cheaper settings
// were measured and rejected (2: operands rounded to fp16, -29 % time, p99 pdf error 4e-4 .. 7e-3; 1: single product,
// 7e-4 .. 8e-3 — the 2x2 determinant of a sharp lobe cancels heavily); the ablation builds live in git history
// (commit 5a33ef0), not in the product source.
constexpr int kTangentPrec = 3;
constexpr int CLK_SLOTS = 256;  // counter pairs of the in-kernel clock measurement, 8 u64 (one 64-B line) apart
constexpr int MAX_SEG = 64;  // materials per segmented launch (the descriptors travel in the kernel arguments)
enum { IO_OPERATOR = 0, IO_PLUGIN = 1, IO_PLUGIN_FULLSPHERE = 2 };

struct ImgLayout {  // byte offsets into the weight image (identical in global memory and LDS)
    int win, wc, wh, wh_lo, wo, wf, wf_lo, wg, wg_lo, bw1, bb1, bw2, bb2, wcs, wcs_lo, wt0, total;
};

struct KParams {
    const char* img;
    ImgLayout L;
    const float* in_a;   // operator: omega_i [N,2]   plugin: wi [N,3]
    const float* in_b;   // sample: x0 [N,2] or null  pdf: omega_o [N,2] / wo [N,3]
    float* out_x;        // sample: x [N,2] / wo [N,3]
    float* out_pdf;      // [N]
    long long N;
    int T;
    int n_hidden;
    int op;
    int io;
    unsigned long long seed, offset;
    // multi-material ("segmented") launch: the query arrays hold nseg contiguous buckets, one per
    // material; workgroups [blk_begin, blk_end) of the grid serve bucket [q_begin, q_end) with that
    // material's weight image.  nseg == 0: ordinary single-material launch over [0, N).
    const float* in_c;   // OP_SAMPLE_PDF: wl [N,3], the direction whose pdf is asked
    float* out_pdf2;     // OP_SAMPLE_PDF: pdf(wi, wl) [N]
    // per-query context (everything derived from wi alone: conditioning term of layer 1 + base-net outputs), see
    // bsdfd_context_bytes: written by whichever sample / pdf launch sees the wi array first (ctx_out), read instead of
    // recomputed by the later ones (ctx_in)
    float* ctx_out;
    const float* ctx_in;
    // sample: Philox counter of row i = offset + rng_index[i] (NULL: offset + i).  A bucketed wavefront passes the rows'
    // ORIGINAL lane indices, so the draws do not depend on the bucketing, the sharding or the GPU count
    const long long* rng_index;
    // plugin io: row i of the launch reads its inputs (wi, x0, wo / wl) from row row_index[i] of the callers' arrays and writes its
    // outputs there (NULL: row i).  A wavefront bucketed by material passes the bucket permutation: the gather of the inputs and the
    // scatter of the results happen in the flow kernel's own loads and stores (bsdfd_opts.row_index).  The Philox counter follows
    // (offset + row_index[i]) unless rng_index is given; segments and the per-query context stay indexed by i.
    const long long* row_index;
    // profiling only (else NULL): every wave adds its lifetime in shader cycles (s_memtime) and in ticks of the constant-rate
    // wall clock (s_memrealtime) to one of CLK_SLOTS counter pairs (one 64-B line each: 16 Ki same-address atomics per launch
    // cost a 0.25 ms launch 6 %); the ratio of the sums is the shader clock the kernel ran at (bsdfd_profile_clock_mhz).
    // Both counters are read by the same wave, so per-CU counter offsets cancel.
    unsigned long long* clk;
    int seg_base;    // segmented launches: buckets served by EARLIER launches of the same call (context slot numbering)
    int chunk_log2;  // a wave takes 2^chunk_log2 consecutive-ish tiles per chunk (see the tile map in the kernel)
    int nseg;
    struct Seg {
        const char* img;
        long long q_begin, q_end;
        int blk_begin, blk_end;
        int chunk_log2, pad;
    } seg[MAX_SEG];
};

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// SiLU and its derivative from one sigmoid.  Every layer's weights are packed so that the MFMA
// delivers zs = -log2(e) * z ("scaled pre-activation"): then sigma(z) = 1 / (1 + 2^zs) needs no
// multiply in front of v_exp_f32, the layer's outputs are hs = zs * s = -log2(e) * silu(z) and
// ts = zts * g = -log2(e) * t, and the NEXT layer's unscaled weights applied to (hs, ts) again
// deliver scaled pre-activations.  The first layer's weights carry the factor -log2(e), the
// output layer's carry -ln 2 (host packing, build_image).  silu'(z) = s + silu(z) (1 - s)
// = fma(hs, -ln2 (1 - s), s).
// v_exp_f32 / v_rcp_f32 are 1-ulp hardware transcendentals (quarter rate: the two of them are
// ~15 of the ~55 VALU cycles a hidden unit costs per step).
constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;
__device__ __forceinline__ void silu_grad_scaled(float zs, float& hs, float& g) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zs));
    hs = zs * s;
    g = fmaf(hs, fmaf(s, kLn2, -kLn2), s);
}
__device__ __forceinline__ float silu(float z) {  // base net: unscaled
    return z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -kLog2e));
}

// hi/lo split of an fp32 value into two fp16-representable fp32 values (see header comment)
__device__ __forceinline__ float hi_part(float x) { return __uint_as_float(__float_as_uint(x) & 0xFFFFE000u); }
__device__ __forceinline__ float hi_part_rn(float x) { return (float)(_Float16)x; }   // (BSDFD_SPLIT_RN: the layer-1 state operands)

// sin and cos of a bounded argument (|a| <~ 1e3; the encoder's arguments are 2^b y with |y| <= pi, b <= 4): Cody-Waite
// reduction by pi/2 in four parts (8 + 11 + 11 bits + remainder: k * part is exact, so is the first subtraction) and
// the Cephes single-precision kernels on [-pi/4, pi/4]; max abs error 9.2e-8 (numpy prototype vs fp64, 8 M
// arguments in [-100, 100]) against 6.9e-8 of a correctly rounded fp32 sin.  ~24 VALU instead of the ~3x longer
// general-argument sincosf (whose Payne-Hanek branch these arguments never take).
__device__ __forceinline__ void sincos_bounded(float a, float& s_out, float& c_out) {
    const float k = rintf(a * 0.6366197466850281f);
    float r = fmaf(k, -1.5703125f, a);
    r = fmaf(k, -0.0004837512969970703f, r);
    r = fmaf(k, -7.549533620476723e-08f, r);
    r = fmaf(k, -2.5633440682570896e-12f, r);
    const float z = r * r;
    float p = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    p = fmaf(p, z, -1.6666654611e-1f);
    const float s = fmaf(p * z, r, r);
    float q = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    q = fmaf(q, z, 4.166664568298827e-2f);
    const float c = fmaf(q * z, z, fmaf(z, -0.5f, 1.0f));
    const int n = (int)k;
    const float ss = (n & 1) ? c : s, cc = (n & 1) ? s : c;
    s_out = __uint_as_float(__float_as_uint(ss) ^ ((unsigned)(n & 2) << 30));
    c_out = __uint_as_float(__float_as_uint(cc) ^ ((unsigned)((n + 1) & 2) << 30));
}

// The encoder's and the spherical net's sines / cosines: the bounded-argument kernel above, libm's sincosf only for
// arguments the reference never produces.  Within-run A/B against sincosf everywhere (profiles/r02_ab/ab4): spherical
// kernels -1.0 %, 64-wide -0.8 %, disk +-0 (noise); p99 pdf error unchanged on all seven golden sets.
__device__ __forceinline__ void sincos_enc(float a, float& s, float& c) {
    if (__builtin_expect(fabsf(a) <= 1024.0f, 1)) sincos_bounded(a, s, c);
    else sincosf(a, &s, &c);
}

// LDS reads the compiler does not schedule or wait for: issued a phase ahead of their use and waited for just before the
// first MFMA that consumes them (cdna_hip_programming.md §5.7, form (ii)).  hipcc itself places a ds_read right in front of
// its use.  Until the wait statement the destination registers hold stale data although the compiler considers them defined:
// whether the code in between leaves them alone depends on the toolchain's register allocation, so the BUILD checks it —
// _lib.build() runs bsdf_diffusion_sampling_amd/_asmcheck.py on the assembly of the compilation it is about to ship and,
// if any instruction touches a pending destination, recompiles this file with -DBSDFD_NO_ASYNC_LDS: the same reads as
// ordinary loads the compiler schedules and waits for itself (~2 % slower, the round-3 `ab2_mim_first` form).
#ifdef BSDFD_NO_ASYNC_LDS
#define BSDFD_LDS_VARIANT "compiler-managed LDS reads (fallback build)"
template <int OFF>
__device__ __forceinline__ void lds_read_b128_async_at(f16x8& dst, unsigned lane_base) {
    typedef const f16x8 __attribute__((address_space(3))) * lds_frag_ptr;
    dst = *reinterpret_cast<lds_frag_ptr>(static_cast<uintptr_t>(lane_base + (unsigned)OFF));
}
#define BSDFD_WAIT2(after, a, b) ((void)0)
#define BSDFD_WAIT4(after, a, b, c, d) ((void)0)
#define BSDFD_WAIT5(after, a, b, c, d, e) ((void)0)
#define BSDFD_WAIT6(after, a, b, c, d, e, f) ((void)0)
#else
#define BSDFD_LDS_VARIANT "asynchronous LDS reads"
template <int OFF>
__device__ __forceinline__ void lds_read_b128_async_at(f16x8& dst, unsigned lane_base) {  // address = lane_base + OFF (immediate)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lane_base), "n"(OFF) : "memory");
}
// One wait statement names every destination of a batch as "+v": all consumers are ordered behind it.  `after` is a value
// the wait is made to depend on as well (the B fragment the activation math of the layer produces): without it the compiler
// schedules the wait — which depends on nothing else — right behind the reads, in front of the math that is meant to hide them.
#define BSDFD_WAIT2(after, a, b) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(after), "+v"(a), "+v"(b) : : "memory")
#define BSDFD_WAIT4(after, a, b, c, d) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(after), "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory")
#define BSDFD_WAIT5(after, a, b, c, d, e) \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(after), "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : : "memory")
#define BSDFD_WAIT6(after, a, b, c, d, e, f) \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(after), "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : : "memory")
#endif
// hipcc's hazard recognizer pads the wait states between an MFMA and the first reader of its result (8 for the 4-pass
// v_mfma_f32_16x16x32_f16, 10 for the 8-pass fp32 shapes: bsdf_diffusion_sampling_amd/_asmcheck.py), but where the MFMAs sit at the
// BOTTOM of a run-time loop and the reader at its head it was found ONE state short on the path through the back edge (round 4,
// every run-time-depth instantiation: 7 of 8 resp. 9 of 10; profiles/r04_ab/mfma_hazard_compiler_gap.txt).  One asm statement at
// the loop head with two wait states inside that names the loop-carried MFMA results as operands, so that none of their readers
// can be scheduled in front of it; _asmcheck verifies the shipped assembly with the strict numbers.
template <int NM, bool JAC>
__device__ __forceinline__ void loop_head_pad(f32x4 (&z)[NM], f32x4 (&a)[NM], f32x4 (&b)[NM]) {
    static_assert(NM == 2 || NM == 4, "one operand list per width");
    if constexpr (NM == 2 && JAC)
        asm volatile("s_nop 1" : "+v"(z[0]), "+v"(z[1]), "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
    else if constexpr (NM == 2)
        asm volatile("s_nop 1" : "+v"(z[0]), "+v"(z[1]));
    else if constexpr (JAC)
        asm volatile("s_nop 1" : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]),
                     "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
    else
        asm volatile("s_nop 1" : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]));
}
union Frag {  // one MFMA B fragment: the lane's 8 K-values of a chunk
    f16x8 v;
    f16x2 p[4];
};
#ifndef BSDFD_SPLIT_RN
#define BSDFD_SPLIT_RN 0   // A/B knob (round 6): hi = x ROUNDED to fp16 (v_cvt_pk_f16_f32, then v_cvt_f32_f16 back) instead of truncated
#endif
template <bool SPLIT>
__device__ __forceinline__ void split_pack(const float (&x)[4], f16x2& h01, f16x2& h23, f16x2& l01, f16x2& l23) {
    if (SPLIT && BSDFD_SPLIT_RN) {
        h01 = (f16x2){(_Float16)x[0], (_Float16)x[1]};
        h23 = (f16x2){(_Float16)x[2], (_Float16)x[3]};
        l01 = (f16x2){(_Float16)(x[0] - (float)h01[0]), (_Float16)(x[1] - (float)h01[1])};
        l23 = (f16x2){(_Float16)(x[2] - (float)h23[0]), (_Float16)(x[3] - (float)h23[1])};
    } else if (SPLIT) {
        const float h0 = hi_part(x[0]), h1 = hi_part(x[1]), h2 = hi_part(x[2]), h3 = hi_part(x[3]);
        h01 = (f16x2){(_Float16)h0, (_Float16)h1};
        h23 = (f16x2){(_Float16)h2, (_Float16)h3};
        l01 = (f16x2){(_Float16)(x[0] - h0), (_Float16)(x[1] - h1)};
        l23 = (f16x2){(_Float16)(x[2] - h2), (_Float16)(x[3] - h3)};
    } else {
        h01 = (f16x2){(_Float16)x[0], (_Float16)x[1]};
        h23 = (f16x2){(_Float16)x[2], (_Float16)x[3]};
    }
}

// ---- precision f16 (single fp16 products, samples only): the sigmoids in PACKED fp16 ----
// A transcendental of both halves of four packed-fp16 registers: the low halves by the plain 16-bit form (gfx9 keeps the
// destination's high half), the high halves by SDWA word selects (the compiler converts, evaluates and re-packs value by value:
// v_cvt_f16_f32 + v_pack_b32_f16, ~1.3x the VALU).  Every result is written >= 3 instructions before it is touched again and one
// wait state separates the block from its consumers: a VALU reading a transcendental's or a dst_sel write's result needs one on
// gfx940-class parts, and the compiler's hazard recogniser does not look inside an asm statement.
#ifndef BSDFD_NO_SDWA_PACK
#define BSDFD_SIGMOID_VARIANT ""
#define BSDFD_PK4_TRANS(OP, R, X)                                                                                             \
    asm(OP "_e32 %0, %4\n\t" OP "_e32 %1, %5\n\t" OP "_e32 %2, %6\n\t" OP "_e32 %3, %7\n\t"                                \
        OP "_sdwa %0, %4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"                                       \
        OP "_sdwa %1, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"                                       \
        OP "_sdwa %2, %6 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"                                       \
        OP "_sdwa %3, %7 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\ts_nop 0"                                \
        : "=&v"(R[0]), "=&v"(R[1]), "=&v"(R[2]), "=&v"(R[3]) : "v"(X[0]), "v"(X[1]), "v"(X[2]), "v"(X[3]))
// eight scaled pre-activations (fp32 accumulators) -> the B fragment of one K = 16 chunk: hs = zs / (1 + 2^zs) evaluated on the
// pre-activation rounded to fp16.  3 plain VALU + 4 transcendentals per pair of units against 4.5 + 4 in fp32; the error is that
// of the fp16 operand the MFMA takes anyway (|x - oracle| p99 3.6e-3 either way on the 64 x 6 teacher at T = 128, contract 2e-2).
// Overflow: 2^zs = inf from zs = 16 -> sigma = 0, hs = 0 (the limit); no 0 x inf (zs is finite in fp16 for |z| < 4.5e4).
__device__ __forceinline__ void act_pack8(const float (&zs)[8], Frag& b) {
    f16x2 zh[4], e[4], a[4], sg[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) zh[k] = (f16x2){(_Float16)zs[2 * k], (_Float16)zs[2 * k + 1]};
    BSDFD_PK4_TRANS("v_exp_f16", e, zh);
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = e[k] + (f16x2){(_Float16)1.0f, (_Float16)1.0f};
    BSDFD_PK4_TRANS("v_rcp_f16", sg, a);
#pragma unroll
    for (int k = 0; k < 4; ++k) b.p[k] = zh[k] * sg[k];
}
#else
// -DBSDFD_NO_SDWA_PACK (the compiler-only build, BSDFD_COMPILER_ONLY_BUILD=1 in _lib.build()): the same arithmetic — pre-activation
// rounded to fp16, v_exp_f16, add, v_rcp_f16, multiply — written value by value for the compiler, which converts and re-packs each
// value itself (~1.3x the VALU of the hand-packed form, +4 % teacher time when measured in round 5) and pads its own hazards.
#define BSDFD_SIGMOID_VARIANT "; compiler-written fp16 sigmoids"
__device__ __forceinline__ void act_pack8(const float (&zs)[8], Frag& b) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const _Float16 z0 = (_Float16)zs[2 * k], z1 = (_Float16)zs[2 * k + 1];
        const _Float16 s0 = __builtin_amdgcn_rcph((_Float16)1.0f + __builtin_elementwise_exp2(z0));
        const _Float16 s1 = __builtin_amdgcn_rcph((_Float16)1.0f + __builtin_elementwise_exp2(z1));
        b.p[k] = (f16x2){z0 * s0, z1 * s1};
    }
}
#endif
__device__ __forceinline__ float sel4(int g, float a0, float a1, float a2, float a3) {
    return g == 0 ? a0 : (g == 1 ? a1 : (g == 2 ? a2 : a3));
}

// Scalar functions of the per-query prologue / epilogue on the hardware transcendentals (round 4: libm's versions are a measurable
// part of the ~100 us a 1 Mi-query spherical launch spends outside its Euler steps, tools/fixed_cost.py).  v_log_f32 / v_exp_f32 are
// 1-ulp log2 / exp2: ln x = ln 2 * log2 x carries <= 2 ulp + 1e-7 |ln x|, far inside the 1e-4 contract (the density tests hold the
// results to 2e-5 of the fp64 oracle).
__device__ __forceinline__ float fast_log(float x) { return kLn2 * __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }
// acos on [-1, 1] for the von Mises SAMPLER only (Abramowitz & Stegun 4.4.46, |error| <= 2e-8 + the hardware sqrt's ulp): the
// value is the sample itself — its density is evaluated at whatever comes out — so 1e-7 of absolute error is immaterial there.
// (cart_to_spher keeps libm's acosf / atan2f: their error is amplified by the encoder's 2^4 and by the flow.)
__device__ __forceinline__ float fast_acos(float x) {
#pragma clang fp contract(off)
    const float a = fabsf(x);
    float p = -0.0012624911f;
    p = fmaf(p, a, 0.0066700901f); p = fmaf(p, a, -0.0170881256f); p = fmaf(p, a, 0.0308918810f);
    p = fmaf(p, a, -0.0501743046f); p = fmaf(p, a, 0.0889789874f); p = fmaf(p, a, -0.2145988016f);
    p = fmaf(p, a, 1.5707963050f);
    const float r = __builtin_amdgcn_sqrtf(fmaxf(1.0f - a, 0.0f)) * p;
    return x < 0.0f ? 3.14159265358979323846f - r : r;
}

// cart_to_spher of the spherical plugins (rendering/brdf_measured_spherical.py:35-39): theta = acos(z / (r + 1e-8)), phi = atan2(y, x).
// Evaluated as written, fp32 loses theta near the pole: the quotient is rounded to a multiple of 6e-8 and acos amplifies that by
// 1 / sin(theta) — up to 6.5e-5 rad on the golden fixtures, which the encoder's 2^4 and the flow turn into 1.0e-4 (p99) of pdf error
// on chm_orange: the reference's own fp32-vs-fp64 distance at plugin level.  The SAME angle in a well-conditioned form: with
// r' = r + eps, cos(theta) = z / r' and sin(theta) = sqrt(r'^2 - z^2) / r' = sqrt(x^2 + y^2 + 2 r eps + eps^2) / r' (a sum of
// non-negative terms), so theta = atan2(sqrt(x^2 + y^2 + 2 r eps + eps^2), z) — identical in real arithmetic, eps included (at
// the pole both give sqrt(2 eps)), 1.5e-7 rad from the fp64 value on the same fixtures.  `ref_pole`: where the reference's fp32
// quotient is not inside (-1, 1) its theta is 0, pi or NaN and its sin(theta) > 5e-5 guard zeroes the density: pdf() keeps that
// decision (rendering/brdf_measured_spherical.py:134).
//
// Both angles are an atan2f of different arguments, and every lane of a query would evaluate them redundantly (~50 VALU
// instructions each; a pdf() launch needs four: theta and phi of wi and of wo).  So the four lanes of a query take ONE angle
// each - lane (g, q) evaluates job g & (NJ - 1) - and the results are handed round with ds_bpermute (the LDS crossbar, not
// the VALU): bit-identical to the redundant evaluation, 150 VALU instructions per tile less in a spherical pdf() launch, 50 in a
// sample() launch (tools/pro_count.sh counted 557 resp. 740 outside the Euler loop before).
struct SphArgs {        // theta = atan2(s, z), phi = atan2(y, x)
    float s, z, y, x;
    bool ref_pole;
#ifdef BSDFD_DIAG_ACOS_AS_WRITTEN
    float theta_ref;    // DIAGNOSTIC BUILD ONLY (tools/acos_diag.py): acosf(z / (r + 1e-8f)), the reference's line as written in fp32
#endif
};
__device__ __forceinline__ SphArgs spher_args(float x, float y, float z) {
    const float eps = 1e-8f;
    const float s2 = x * x + y * y;
    const float r = sqrtf(s2 + z * z);
    SphArgs a;
    a.s = sqrtf(s2 + (2.0f * r * eps + eps * eps));
    a.z = z; a.y = y; a.x = x;
    a.ref_pole = !(fabsf(z / (r + eps)) < 1.0f);
#ifdef BSDFD_DIAG_ACOS_AS_WRITTEN
    a.theta_ref = acosf(z / (r + eps));
#endif
    return a;
}
template <int NJ>   // NJ = 2 or 4 jobs (Y[k], X[k]); out[k] = atan2f(Y[k], X[k]) in every lane of the query
__device__ __forceinline__ void atan2_by_lane(const float (&Y)[NJ], const float (&X)[NJ], int g, int q, float (&out)[NJ]) {
    static_assert(NJ == 2 || NJ == 4, "jobs are dealt to the 4 lanes of a query");
    const int j = g & (NJ - 1);
    float yy = Y[0], xx = X[0];
#pragma unroll
    for (int k = 1; k < NJ; ++k) {
        yy = j == k ? Y[k] : yy;
        xx = j == k ? X[k] : xx;
    }
    const float a = atan2f(yy, xx);
#pragma unroll
    for (int k = 0; k < NJ; ++k) out[k] = __shfl(a, 16 * k + q, 64);   // row k of the wave holds job k
}

// log I0(kappa): the two polynomials of torch.distributions.von_mises._log_modified_bessel_fn
// (torch 2.10; call site rendering/utils/model.py:314), split at 3.75.
__device__ __forceinline__ float log_i0(float k) {
    if (k < 3.75f) {
        float y = k * (1.0f / 3.75f);
        y = y * y;
        float p = 0.0045813f;
        p = fmaf(p, y, 0.0360768f); p = fmaf(p, y, 0.2659732f); p = fmaf(p, y, 1.2067492f);
        p = fmaf(p, y, 3.0899424f); p = fmaf(p, y, 3.5156229f); p = fmaf(p, y, 1.0f);
        return fast_log(p);
    }
    const float y = 3.75f * __builtin_amdgcn_rcpf(k);
    float p = 0.00392377f;
    p = fmaf(p, y, -0.01647633f); p = fmaf(p, y, 0.02635537f); p = fmaf(p, y, -0.02057706f);
    p = fmaf(p, y, 0.00916281f); p = fmaf(p, y, -0.00157565f); p = fmaf(p, y, 0.00225319f);
    p = fmaf(p, y, 0.01328592f); p = fmaf(p, y, 0.39894228f);
    return k - 0.5f * fast_log(k) + fast_log(p);
}

__device__ __forceinline__ float softplus(float x) {  // nn.Softplus(beta=1, threshold=20)
    // (1 + e^x loses e^x's low bits for x << 0: an ABSOLUTE error <= 6e-8 on kappa = softplus + 1e-3, i.e. <= 1.2e-7 on log p)
    return x > 20.0f ? x : fast_log(1.0f + fast_exp(x));
}

// Best & Fisher rejection sampler for VonMises(mu, kappa)
// (torch/distributions/von_mises.py::_rejection_sample; call site rendering/utils/model.py:305).
// * Proposal constant: torch forms rho = (tau - sqrt(2 tau)) / (2 kappa) in fp64 because the difference
//   cancels in fp32; here the rationalised form rho = 2 kappa / (tau + sqrt(2 tau)) (tau (tau - 2) =
//   4 kappa^2) has no cancellation and is evaluated in fp32.  r only shapes the envelope — the
//   accept test uses the same r, so the sampler is exact for any r > 1.
// * The 4 lanes of a query (lane = 16 g + q) test 4 CONSECUTIVE proposals of the query's Philox
//   stream at once (proposal index 4 round + g); the first accepted one in stream order wins, so the
//   draw equals the sequential loop's while a wave needs ~1.2 rounds instead of ~3.5 (max over its
//   16 queries of a geometric trip count with acceptance >= 0.66).
__device__ __forceinline__ float von_mises_sample(float mu, float kappa, unsigned k0, unsigned k1, unsigned q_lo,
                                                  unsigned q_hi, int lane) {
    // No implicit fma contraction in here: the accept test is a comparison, and a product fused in one kernel instantiation but
    // not in another would flip it for the occasional query — every instantiation (single-op, fused, segmented) must draw the
    // same sample for the same Philox counter.
#pragma clang fp contract(off)
    float r;
    // (hardware rcp / sqrt / log / cos and the polynomial acos: r only shapes the envelope, the accept test is a comparison of
    //  random numbers, and the accepted angle is the sample itself — ulp-level differences change nothing statistically)
    if (kappa < 1e-5f) {
        r = __builtin_amdgcn_rcpf(kappa) + kappa;
    } else {
        const float tau = 1.0f + __builtin_amdgcn_sqrtf(1.0f + 4.0f * kappa * kappa);
        const float rho = 2.0f * kappa * __builtin_amdgcn_rcpf(tau + __builtin_amdgcn_sqrtf(2.0f * tau));
        r = (1.0f + rho * rho) * __builtin_amdgcn_rcpf(2.0f * rho);
    }
    const int g = lane >> 4, q = lane & 15;
    float x = 0.0f;
    bool done = false;
    for (unsigned round = 0; round < 64u; ++round) {
        unsigned u[4];
        philox4x32(k0, k1, q_lo, q_hi, round * 4u + (unsigned)g + 1u, 0x564d6973u, u);  // "VMis"
        const float u1 = u01_open(u[0]), u2 = u01_open(u[1]), u3 = u01_open(u[2]);
        const float z = __builtin_amdgcn_cosf(0.5f * u1);   // v_cos_f32 takes revolutions: cos(pi u1)
        const float f = (1.0f + r * z) * __builtin_amdgcn_rcpf(r + z);
        const float c = kappa * (r - f);
        const bool accept = (c * (2.0f - c) - u2 > 0.0f) || (fast_log(c * __builtin_amdgcn_rcpf(u2)) + 1.0f - c >= 0.0f);
        const float a = fast_acos(fminf(fmaxf(f, -1.0f), 1.0f));
        const float cand = (u3 - 0.5f) < 0.0f ? -a : a;
        const unsigned long long acc_mask = __builtin_amdgcn_ballot_w64(accept);
        const unsigned long long mine = (acc_mask >> q) & 0x0001000100010001ull;  // bit 16 g' = lane (g', q)
        const int first_g = mine ? (__builtin_ctzll(mine) >> 4) : 0;
        const float got = __shfl(cand, first_g * 16 + q, 64);
        if (!done && mine) { x = got; done = true; }
        if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
    }
    const float two_pi = 6.28318530717958647692f, pi = 3.14159265358979323846f;
    const float t = x + pi + mu;
    float w = fmaf(-two_pi, floorf(t * (1.0f / two_pi)), t);   // t mod 2 pi, in [0, 2 pi) up to rounding
    if (w < 0.0f) w += two_pi;
    if (w >= two_pi) w -= two_pi;
    return w - pi;
}

}  // namespace
