// flow32.hip — the flow sampler on 32-QUERY TILES: the two nets the reference's plugins load (disk 25-32x3-2,
// rendering/utils/model.py:479-501; spherical 26-32x4-2, :422-446; split-fp16 arithmetic) on v_mfma_f32_32x32x16_f16.
//
// Why a second tiling (profiles/HISTORY.md, round 5): in the VALU-heavy instruction stream of the Euler step a v_mfma_f32_16x16x32_f16 hides
// NO other work (its 16 matrix-pipe cycles add to the VALU time), a 32x32x16 lets ~9 cycles of the co-resident waves' VALU work
// through per instruction (tools/ubench/mfma_src, profiles/r04_ab/mfma_shapes_ubench.txt).  The 32-wide nets fit the shape exactly:
// M = 32 = all hidden units, N = 32 queries, two K = 16 chunks per contraction — no padded rows in the hidden layers.
//
// Mapping (compare bsdfd.hip's header):
//   * a wave64 owns 32 queries; lane = (h = lane >> 5, n = lane & 31) holds, for query n, the 16 hidden units
//     u(v, h) = 8 (v >> 2) + 4 h + (v & 3), v = 0..15 — the C/D layout of the 32x32 shapes.  The K index of the next contraction
//     is permuted to match (chunk c, lane half h, slot j  <->  unit u(8 c + j, h); the host packs the A fragments accordingly),
//     so the lane's accumulator registers 8c .. 8c+7, split hi/lo and packed, ARE the B fragment of chunk c: activations never
//     leave registers between layers.
//   * the query's state is DISTRIBUTED over its two lanes: lane h = 0 carries x0 (theta), lane h = 1 carries x1 (phi).  Layer 1
//     is ONE fp16 MFMA per step: the B slots of a lane are [v_hi, v_lo, v_hi, v_lo, w_hi, w_lo, w_hi, w_lo] with (v, w) = (x0, alpha) |
//     (x1, 0) (disk) or (theta, alpha) | (sin phi, cos phi) (spherical), the A slots [W_hi, W_hi, W_lo, W_lo] of the matching columns
//     of W1 (all four products of the two-way splits inside one instruction; the hi parts of these state operands are ROUNDED to
//     fp16, BSDFD_T32_L1_RN — every other split truncates), C-in = the per-query conditioning term.
//   * the two-row output layer is an fp32 VALU dot over the lane's 16 units (the last hidden activation is never split) and one
//     v_permlane32_swap, which leaves v0 in the lower and v1 in the upper half-wave: exactly where x0 and x1 live.
//   * the Jacobian meets in the middle as in the 16-query kernels (MIM / MIMS); its reduction runs over the 2 lanes of a query
//     (4 swaps per 32 queries instead of 5 per 16).
//   * per-tile prologue: lane h encodes dimension h of omega_i (5 sincos), the conditioning term is 6 split-fp16 MFMAs (disk) or
//     11 exact-fp32 v_mfma_f32_32x32x2_f32 (spherical), the base net's first layer 4 split-fp16 MFMAs (all four products), its
//     second layer fp32 VALU.
//   * 3 waves per SIMD (2 for the fused spherical kernel); no kernel uses scratch: the lane number is re-derived per tile and the
//     query's row after the Euler loop instead of being carried across it.
// Same operators, same plugin variants, same context / rng_index / segmented-launch semantics as the 16-query kernels; the base
// draws are bit-identical to theirs (same Philox counters, same arithmetic).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "bsdfd.h"
#include "common.h"
#include "flow_dev.h"
#include "flow32.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int FR32 = 64 * 16;  // bytes of one A fragment (32 rows x 16 K-values of fp16)

// build knobs of the A/B runs (tools/ab_build32.sh); the defaults are the product
#ifndef BSDFD_T32_WAVES
#define BSDFD_T32_WAVES 3        // waves per SIMD the register allocator is asked to make room for (168 VGPRs)
#endif
#ifndef BSDFD_T32_CACC_LDS
#define BSDFD_T32_CACC_LDS 2     // the per-query conditioning term in a per-wave LDS slab instead of 16 VGPRs: 0 never, 1 always,
#endif                           // 2 = the spherical and the fused kernels (they spill at 3 waves/SIMD otherwise; the disk single-op kernel fits)
#ifndef BSDFD_T32_BASE_SPLIT
#define BSDFD_T32_BASE_SPLIT 1   // first layer of the base net as 4 split-fp16 MFMAs (all four products of the two-way splits) instead of 7
                                 // exact-fp32 ones: -2.5 % kernel time at the plugin's T = 4, worst of the 77 shipped sets 3.9e-5 -> 4.0e-5
                                 // (sample p99) / 6.3e-5 -> 6.3e-5 (pdf), profiles/r05_ab/ab32_base_net_split*
#endif
#ifndef BSDFD_T32_FUSED_SPH_WAVES
#define BSDFD_T32_FUSED_SPH_WAVES 2   // the fused spherical sample+pdf kernel keeps more state across its two Euler loops
#endif
#ifndef BSDFD_T32_SPH_COND_SPLIT
#define BSDFD_T32_SPH_COND_SPLIT 0   // the SPHERICAL conditioning term as split-fp16 MFMAs with all FOUR products of the two-way splits (8 MFMAs)
#endif                               // instead of 11 exact-fp32 v_mfma_f32_32x32x2_f32.  Round 6, NEGATIVE: -0.5 % kernel time (T = 8) for 6-10 % of
                                     // the accuracy margin (worst of the 50 spherical-domain sets at 65 536 queries: sample 5.9e-5 -> 6.5e-5, pdf at
                                     // produced directions 6.7e-5 -> 7.2e-5, at fresh ones 8.6e-5 -> 9.1e-5 [9.8e-5]; profiles/r06_ab/ab32_spherical_*)
#ifndef BSDFD_T32_L1_RN
#define BSDFD_T32_L1_RN 1             // the hi parts of layer 1's STATE operands (x / theta, sin phi, cos phi, alpha) are ROUNDED to fp16, not
#endif                                // truncated: hi + lo then carries 23 bits instead of 22 for one v_cvt_f32_f16 in place of a v_and.  The state
                                      // operands' quantisation was the largest single term of the worst (set, call) of the 77-set sweep (CPU:
                                      // tools/archive/r06_state_split_diag.py, bsdf_24 pdf() at fresh directions 5.2e-5 -> 3.2e-5 of its 8.6e-5);
                                      // measured: 8.6e-5 [9.2e-5] -> 7.6e-5 [8.3e-5], time and J/query 0.999-1.002 (profiles/r06_ab/ab32_l1_rn.txt)
#ifndef BSDFD_T32_SPH_FOLD_T0
#define BSDFD_T32_SPH_FOLD_T0 1      // spherical d/dtheta tangent through a folded matrix: the layer-1 tangent is g1 . W1[:, theta] with a CONSTANT
#endif                               // column, so W2 (g1 . W1[:, theta]) = F_theta g1, F_theta = W2 diag(W1[:, theta]) packed by the host (the disk
                                     // kernel's F_i).  Per step 16 multiplies and the read of W1[:, theta] less, the same MFMAs and splits.
                                     // Round 6: -1.1 % kernel time, -1.6 % J/query (T = 8), accuracy unchanged on all 50 sets (same file)
#ifndef BSDFD_T32_JAC2
#define BSDFD_T32_JAC2 0         // A/B knob (round 6; negative, profiles/r06_ab/): the Jacobian-ONLY contractions (folded matrices, spherical
#endif                           // tangent layers) in TWO products — 1: x rounded to fp16, no x_lo terms (and no hi/lo split of those vectors);
                                 // 2: W_lo terms dropped.  0 = three products like the activations (the product)
template <int DOMAIN, bool FUSED>
struct CaccLds {
    static constexpr bool on = BSDFD_T32_CACC_LDS == 1 || (BSDFD_T32_CACC_LDS == 2 && (FUSED || DOMAIN == BSDFD_DOMAIN_SPHERICAL));
    static constexpr int slab = on ? 64 * 64 : 0;   // bytes per wave
};

// byte offsets into the weight image — compile-time constants per domain (the host's build_image32 uses the same struct)
template <int DOMAIN>
struct L32 {
    static constexpr bool SPH = DOMAIN == BSDFD_DOMAIN_SPHERICAL;
    static constexpr int NH = SPH ? 4 : 3;
    static constexpr int A1 = 0;                                   // layer-1 state fragment
    static constexpr int WT0 = A1 + FR32;                          // spherical without BSDFD_T32_SPH_FOLD_T0: W1[:, theta] in accumulator layout, 64 lanes x 16 floats
    static constexpr int WC = WT0 + ((SPH && !BSDFD_T32_SPH_FOLD_T0) ? 64 * 64 : 0);   // conditioning: disk 4 fragments (hi c0, hi c1, lo c0, lo c1); spherical 11 x 64 floats
    static constexpr int WH = WC + ((SPH && !BSDFD_T32_SPH_COND_SPLIT) ? 11 * 256 : 4 * FR32);    // hidden matrices W2 .. W_NH: 4 fragments each
    static constexpr int WF = WH + (NH - 1) * 4 * FR32;            // disk: F0, F1 = W2 diag(W1[:, i]); spherical (BSDFD_T32_SPH_FOLD_T0): F_theta
    static constexpr int WG = WF + (SPH ? (BSDFD_T32_SPH_FOLD_T0 ? 4 * FR32 : 0) : 8 * FR32);   // G0, G1 = W_NH^T diag(Wout[j, :])
    static constexpr int WOUT = WG + 8 * FR32;                     // 2 halves x 16 x (Wout[0][u], Wout[1][u]) floats
    static constexpr int BW1 = WOUT + 256;                         // base net layer 1: 7 x 64 floats (A operands of the 32x32x2 MFMA)
    static constexpr int BB1 = BW1 + 7 * 256;                      // 2 halves x 8 floats
    static constexpr int BW2 = BB1 + 64;                           // 2 halves x 8 units x 4 outputs
    static constexpr int BB2 = BW2 + 256;                          // 4 floats
    static constexpr int BW1S = BB2 + 16;                          // base net layer 1 as fp16 A fragments (hi, lo): BSDFD_T32_BASE_SPLIT
    static constexpr int TOTAL = BW1S + 2 * FR32;
};

__host__ __device__ constexpr int unit32(int v, int h) { return 8 * (v >> 2) + 4 * h + (v & 3); }

__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32f(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// v_permlane32_swap(x, y): x[32..63] <-> y[0..31].  Afterwards x + y is, in the lower half-wave, x summed over the lane pair
// (l, l + 32) and, in the upper half, y summed over it.
__device__ __forceinline__ void swap32(float& x, float& y) {
    const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(t[0]);
    y = __uint_as_float(t[1]);
}
// both halves get (lower's value, upper's value) of a register
__device__ __forceinline__ void both32(float x, float& lo, float& up) {
    lo = x; up = x;
    swap32(lo, up);
}

// the lane's 16 values of a vector -> the B fragments (hi, lo) of the two K = 16 chunks of the next contraction
// (SPLIT = false, precision f16: rounded to fp16, no lo part)
template <bool SPLIT = true>
__device__ __forceinline__ void split16(const float (&x)[16], Frag (&hi)[2], Frag (&lo)[2]) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float x4[4] = {x[8 * c + 4 * k], x[8 * c + 4 * k + 1], x[8 * c + 4 * k + 2], x[8 * c + 4 * k + 3]};
            split_pack<SPLIT>(x4, hi[c].p[2 * k], hi[c].p[2 * k + 1], lo[c].p[2 * k], lo[c].p[2 * k + 1]);
        }
}

// a layer's 16 sigmoids: hs = zs sigma, g = silu' (flow_dev.h: silu_grad_scaled); WITH_G = false: hs alone (no Jacobian)
template <bool WITH_G>
__device__ __forceinline__ void act16(const f32x16& z, float (&hs)[16], float (&g)[16]) {
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        if (WITH_G) silu_grad_scaled(z[v], hs[v], g[v]);
        else hs[v] = z[v] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z[v]));
    }
}

// a 32-wide layer (the lane's 16 units) -> the hi fragments of split16's layout
__device__ __forceinline__ void act_pack16(const f32x16& z, Frag (&hi)[2]) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float z8[8] = {z[8 * c], z[8 * c + 1], z[8 * c + 2], z[8 * c + 3], z[8 * c + 4], z[8 * c + 5], z[8 * c + 6], z[8 * c + 7]};
        act_pack8(z8, hi[c]);
    }
}

// one 32 x 32 matrix = 4 fragments (hi chunk 0, hi chunk 1, lo chunk 0, lo chunk 1)
struct Mat32 {
    f16x8 h0, h1, l0, l1;
};
__device__ __forceinline__ Mat32 load_mat(const char* smem, int off, int lane) {
    const f16x8* m = reinterpret_cast<const f16x8*>(smem + off) + lane;
    Mat32 r;
    r.h0 = m[0]; r.h1 = m[64]; r.l0 = m[128]; r.l1 = m[192];
    return r;
}
// (W_hi + W_lo) (x_hi + x_lo) without the lo lo term: hi hi + hi lo + lo hi in ONE fp32 accumulator, 6 MFMAs
// (SPLIT = false, precision f16: the hi hi product alone, 2 MFMAs)
template <bool SPLIT = true>
__device__ __forceinline__ f32x16 mm6(const Mat32& w, const Frag (&xh)[2], const Frag (&xl)[2]) {
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 a = mfma32(w.h0, xh[0].v, zero16);
    a = mfma32(w.h1, xh[1].v, a);
    if (!SPLIT) return a;
    a = mfma32(w.h0, xl[0].v, a);
    a = mfma32(w.h1, xl[1].v, a);
    a = mfma32(w.l0, xh[0].v, a);
    return mfma32(w.l1, xh[1].v, a);
}
// two such products issued term by term, alternating accumulators (matrices wa, wb; vectors xa, xb)
__device__ __forceinline__ void mm6x2(const Mat32& wa, const Frag (&xah)[2], const Frag (&xal)[2], const Mat32& wb, const Frag (&xbh)[2],
                                      const Frag (&xbl)[2], f32x16& a, f32x16& b) {
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    a = mfma32(wa.h0, xah[0].v, zero16); b = mfma32(wb.h0, xbh[0].v, zero16);
    a = mfma32(wa.h1, xah[1].v, a); b = mfma32(wb.h1, xbh[1].v, b);
    if (BSDFD_T32_JAC2 != 1) {
        a = mfma32(wa.h0, xal[0].v, a); b = mfma32(wb.h0, xbl[0].v, b);
        a = mfma32(wa.h1, xal[1].v, a); b = mfma32(wb.h1, xbl[1].v, b);
    }
    if (BSDFD_T32_JAC2 != 2) {
        a = mfma32(wa.l0, xah[0].v, a); b = mfma32(wb.l0, xbh[0].v, b);
        a = mfma32(wa.l1, xah[1].v, a); b = mfma32(wb.l1, xbh[1].v, b);
    }
}
// the hi/lo split of a vector that feeds Jacobian-only contractions (BSDFD_T32_JAC2 = 1: rounded to fp16, no lo part)
__device__ __forceinline__ void split16_jac(const float (&x)[16], Frag (&hi)[2], Frag (&lo)[2]) {
    if (BSDFD_T32_JAC2 == 1) split16<false>(x, hi, lo); else split16<true>(x, hi, lo);
}

// Best & Fisher rejection sampler, 32-query tiles: the 2 lanes of a query test 4 consecutive proposals of the query's Philox
// stream per round (lane h: proposals 4 round + 2 h + {0, 1}); the first accepted one in stream order wins, so the draw equals
// von_mises_sample's (flow_dev.h) and the sequential loop's of torch/distributions/von_mises.py::_rejection_sample.
__device__ __forceinline__ float von_mises_sample32(float mu, float kappa, unsigned k0, unsigned k1, unsigned q_lo, unsigned q_hi,
                                                    int lane) {
#pragma clang fp contract(off)
    float r;
    if (kappa < 1e-5f) {
        r = __builtin_amdgcn_rcpf(kappa) + kappa;
    } else {
        const float tau = 1.0f + __builtin_amdgcn_sqrtf(1.0f + 4.0f * kappa * kappa);
        const float rho = 2.0f * kappa * __builtin_amdgcn_rcpf(tau + __builtin_amdgcn_sqrtf(2.0f * tau));
        r = (1.0f + rho * rho) * __builtin_amdgcn_rcpf(2.0f * rho);
    }
    const int h = lane >> 5, n = lane & 31;
    float x = 0.0f;
    bool done = false;
    for (unsigned round = 0; round < 64u; ++round) {
        bool has = false;
        float mine = 0.0f;
#pragma unroll
        for (unsigned e = 0; e < 2u; ++e) {
            unsigned u[4];
            philox4x32(k0, k1, q_lo, q_hi, round * 4u + 2u * (unsigned)h + e + 1u, 0x564d6973u, u);  // "VMis"
            const float u1 = u01_open(u[0]), u2 = u01_open(u[1]), u3 = u01_open(u[2]);
            const float z = __builtin_amdgcn_cosf(0.5f * u1);
            const float f = (1.0f + r * z) * __builtin_amdgcn_rcpf(r + z);
            const float c = kappa * (r - f);
            const bool accept = (c * (2.0f - c) - u2 > 0.0f) || (fast_log(c * __builtin_amdgcn_rcpf(u2)) + 1.0f - c >= 0.0f);
            const float a = fast_acos(fminf(fmaxf(f, -1.0f), 1.0f));
            const float cand = (u3 - 0.5f) < 0.0f ? -a : a;
            if (!has && accept) { mine = cand; has = true; }
        }
        const unsigned long long acc_mask = __builtin_amdgcn_ballot_w64(has);
        const bool lo_has = (acc_mask >> n) & 1ull, up_has = (acc_mask >> (n + 32)) & 1ull;
        const float got = __shfl(mine, lo_has ? n : n + 32, 64);
        if (!done && (lo_has || up_has)) { x = got; done = true; }
        if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
    }
    const float two_pi = 6.28318530717958647692f, pi = 3.14159265358979323846f;
    const float t = x + pi + mu;
    float w = fmaf(-two_pi, floorf(t * (1.0f / two_pi)), t);
    if (w < 0.0f) w += two_pi;
    if (w >= two_pi) w -= two_pi;
    return w - pi;
}

// ---------------------------------------------------------------------------------------------
// The kernel.  DOMAIN: BSDFD_DOMAIN_*; JAC: track the Jacobian determinant (false: bsdfd_flow_samples_only — the same trajectory
// as the sampling kernel, bit for bit); FUSED: the two-phase OP_SAMPLE_PDF loop (its own instantiation); SPLIT: split-fp16
// contractions (precision split3) — false: single fp16 products (precision f16; instantiated without the Jacobian only: the reflow
// teacher sampling of the 32-wide nets, learning_repo_cleanup/disk_domain_sampling.py:93-110).
// Per lane the step keeps 16-register vectors where the 16-query kernels keep 8; BSDFD_T32_WAVES = 3 waves per SIMD (168 VGPRs).
// ---------------------------------------------------------------------------------------------
template <int DOMAIN, bool JAC, bool FUSED, bool SPLIT>
__global__ __launch_bounds__(256, (FUSED && DOMAIN == BSDFD_DOMAIN_SPHERICAL) ? BSDFD_T32_FUSED_SPH_WAVES : BSDFD_T32_WAVES) void flow_kernel32(const KParams p) {
    static_assert(SPLIT || !JAC, "the single-product (f16) instantiations exist without the Jacobian only");
    using LY = L32<DOMAIN>;
    constexpr bool SPH = DOMAIN == BSDFD_DOMAIN_SPHERICAL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (p.clk) { clk_c0 = __builtin_readcyclecounter(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    const char* img = p.img;
    long long q_begin = 0, q_end = p.N;
    int blk = blockIdx.x, nblk = gridDim.x, cl = p.chunk_log2;
    int sidx = 0;
    if (p.nseg > 0) {
        for (int i = 1; i < p.nseg; ++i)
            if ((int)blockIdx.x >= p.seg[i].blk_begin) sidx = i;
        img = p.seg[sidx].img;
        q_begin = p.seg[sidx].q_begin;
        q_end = p.seg[sidx].q_end;
        blk = blockIdx.x - p.seg[sidx].blk_begin;
        nblk = p.seg[sidx].blk_end - p.seg[sidx].blk_begin;
        cl = p.seg[sidx].chunk_log2;
    }
    {
        const uint4* src = reinterpret_cast<const uint4*>(img);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (int i = threadIdx.x; i < LY::TOTAL / 16; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();

    const int lane0 = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves_per_block = blockDim.x >> 6;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // launch-uniform values the VALU computes (there is no scalar fp64 divide) are moved to SGPRs: as VGPRs they would be held
    // across the whole kernel
    auto uniform_f = [](float x) -> float { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x))); };
    auto uniform_d = [](double x) -> double {
        const unsigned long long b = (unsigned long long)__double_as_longlong(x);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    const double invT_d = uniform_d(1.0 / (double)p.T);
    const bool t_pow2 = (p.T & (p.T - 1)) == 0;
    const float invT = uniform_f((float)invT_d);
    const int nphase = FUSED ? 2 : 1;
    const bool reverse1 = (p.op == OP_PDF);
    const float cstep1 = reverse1 ? -invT : invT;
    const long long ntiles = (q_end - q_begin + 31) / 32;

    // tile -> wave map: as in the 16-query kernels (chunks of 2^cl x waves_per_block consecutive tiles, round-robin over the grid)
    const long long chunk = (long long)waves_per_block << cl;
    for (long long it = 0;; ++it) {
        const long long chunk_base = ((it >> cl) * nblk + blk) * chunk;
        if (chunk_base >= ntiles) break;
        const long long tile = chunk_base + (it & ((1 << cl) - 1)) * waves_per_block + wave;
        if (tile >= ntiles) continue;
        // The lane number is re-derived per tile from an opaque copy: everything computed from it (LDS addresses in five
        // scalings, the half-wave selects) would otherwise be hoisted out of the tile loop and held in registers for the whole
        // kernel — the fused spherical instantiation spilled 19 of them at kernel entry.
        auto opaque = [](int x) -> int { asm volatile("" : "+v"(x)); return x; };
        const int lane = opaque(lane0);
        const int h = lane >> 5;
        const int n = lane & 31;
        // Row of this lane's query (clamped on the ragged last tile).  The epilogue and the fused kernel's phase switch RECOMPUTE it
        // from the wave-uniform tile base and another opaque copy of the lane number instead of keeping the 64-bit row — and every
        // address derived from it — in registers across the Euler loop, where there are none to spare.
        const long long tile_q0 = q_begin + tile * 32;
        auto row_of = [&](int nn, bool& in_range) -> long long {
            const long long r = tile_q0 + nn;
            in_range = r < q_end;
            return in_range ? r : q_end - 1;
        };
        bool in_range0;
        const long long qi = row_of(n, in_range0);
        // row of the callers' arrays this query reads and writes (bsdfd_opts.row_index: a bucketed wavefront hands over the bucket
        // permutation instead of gathered copies); re-read in the epilogue rather than carried across the Euler loop
        auto user_row = [&](long long r) -> long long { return p.row_index ? p.row_index[r] : r; };
        const long long qu = user_row(qi);

        // ---------------- inputs ---------------------------------------------------------------------
        // yh: this lane's coordinate of the condition omega_i (lane h encodes dimension h); xs: this lane's coordinate of the
        // point the flow starts from (pdf: omega_o; sample with an injected base point: x0)
        float yh = 0.f, wi_z = 1.0f;
        float xs = 0.f, wo_z = 1.0f, wo_sin = 1.0f;
        float xo0 = 0.f, xo1 = 0.f;   // both coordinates of an injected / asked point (base density, epilogue)
        bool wo_pole = false;
        float xi0 = 0.f, xi1 = 0.f;   // FUSED: injected x0
        const bool have_ctx = !FUSED && p.ctx_in != nullptr;
        auto load_dir = [&](const float* dir, long long qi) {   // plugin io: the direction whose pdf is asked -> start point of the reverse flow
            const float ox = dir[qi * 3 + 0], oy = dir[qi * 3 + 1], oz = dir[qi * 3 + 2];
            wo_z = oz;
            wo_sin = sqrtf(ox * ox + oy * oy);  // Mitsuba Frame3f::sin_theta
            if (!SPH) {
                xs = h ? oy : ox;
            } else {   // cart_to_spher (flow_dev.h: spher_args): lane h = 0 evaluates theta, lane h = 1 phi — one atan2f each
                const SphArgs ao = spher_args(ox, oy, oz);
                wo_pole = ao.ref_pole;
                xs = atan2f(h ? ao.y : ao.s, h ? ao.x : ao.z);
#ifdef BSDFD_DIAG_ACOS_AS_WRITTEN
                if (!h) xs = ao.theta_ref;
#endif
            }
        };
        if (p.io == IO_OPERATOR) {
            const float2 c2 = reinterpret_cast<const float2*>(p.in_a)[qi];
            yh = h ? c2.y : c2.x;
            if (p.op == OP_PDF || p.in_b != nullptr) {
                const float2 b2 = reinterpret_cast<const float2*>(p.in_b)[qi];
                xs = h ? b2.y : b2.x;
                xo0 = b2.x; xo1 = b2.y;
            }
        } else {
            const float wx = p.in_a[qu * 3 + 0], wy = p.in_a[qu * 3 + 1], wz = p.in_a[qu * 3 + 2];
            wi_z = wz;
            if (!SPH) {
                yh = h ? wy : wx;  // rendering/brdf_measured_disk.py:66-67
            } else if (!have_ctx) {
                const SphArgs ai = spher_args(wx, wy, wz);  // rendering/brdf_measured_spherical.py:35-39
                yh = atan2f(h ? ai.y : ai.s, h ? ai.x : ai.z);
#ifdef BSDFD_DIAG_ACOS_AS_WRITTEN
                if (!h) yh = ai.theta_ref;
#endif
            }
            if (!FUSED && p.op == OP_PDF) load_dir(p.in_b, qu);
            if ((FUSED || p.op != OP_PDF) && p.in_b != nullptr) {  // injected base sample
                const float2 b2 = reinterpret_cast<const float2*>(p.in_b)[qu];
                if (FUSED) { xi0 = b2.x; xi1 = b2.y; } else { xs = h ? b2.y : b2.x; xo0 = b2.x; xo1 = b2.y; }
            }
        }

        f32x16 cacc;
        f32x4* const cslab = reinterpret_cast<f32x4*>(smem + LY::TOTAL + wave * CaccLds<DOMAIN, FUSED>::slab) + lane;   // [k][lane] f32x4
        f32x4 bo;
        constexpr long long CTX_V4 = 4 * 64 + 32;  // f32x4 per tile: cacc (4 per lane) + bo per query
        const long long ctx_slot = ((q_begin + tile * 32) >> 5) + p.seg_base + sidx;
        if (have_ctx) {
            const f32x4* c = reinterpret_cast<const f32x4*>(p.ctx_in) + ctx_slot * CTX_V4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 t4 = c[k * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) cacc[4 * k + r] = t4[r];
            }
            bo = c[4 * 64 + n];
        } else {
            // ---------------- positional encoding: lane h encodes dimension h ----------------------------
            // e[2 b + fn] = fn(2^b y_h), fn = sin, cos; e[10] = y_h  (rendering/utils/model.py:26-57)
            float e[11];
#pragma unroll
            for (int b = 0; b < PE_BANDS; ++b) {
                // (the fused kernel is plugin io only: there a spherical condition is an atan2f result, |2^b y| <= 16 pi, and the
                //  general-argument branch of sincos_enc is dead code — which the compiler cannot know and pays registers for)
                if constexpr (FUSED && SPH) sincos_bounded(yh * (float)(1 << b), e[2 * b], e[2 * b + 1]);
                else sincos_enc(yh * (float)(1 << b), e[2 * b], e[2 * b + 1]);
            }
            e[10] = yh;
            // ---------------- conditioning term c = W1[:, PE] PE(omega_i) -----------------------------
            if (!SPH || BSDFD_T32_SPH_COND_SPLIT) {   // split-fp16: the lane's 11 values are K slots of two K = 16 chunks (the 16-query kernels' SPLIT_PRO)
                const float e16[16] = {e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], e[8], e[9], e[10], 0.f, 0.f, 0.f, 0.f, 0.f};
                Frag eh[2], el[2];
                split16(e16, eh, el);
                const Mat32 wc = load_mat(smem, LY::WC, lane);
                cacc = mfma32(wc.h0, eh[0].v, zero16);
                cacc = mfma32(wc.h1, eh[1].v, cacc);
                cacc = mfma32(wc.h0, el[0].v, cacc);
                cacc = mfma32(wc.h1, el[1].v, cacc);
                cacc = mfma32(wc.l0, eh[0].v, cacc);
                cacc = mfma32(wc.l1, eh[1].v, cacc);
                if (SPH) {   // the spherical term keeps the fourth product too (lo . lo)
                    cacc = mfma32(wc.l0, el[0].v, cacc);
                    cacc = mfma32(wc.l1, el[1].v, cacc);
                }
            } else {      // exact fp32 chains (the spherical nets keep them: bsdfd.hip, SPLIT_PRO)
                const float* Lwc = reinterpret_cast<const float*>(smem + LY::WC);
                cacc = zero16;
#pragma unroll
                for (int j = 0; j < 11; ++j) cacc = mfma32f(Lwc[j * 64 + lane], e[j], cacc);
            }
            // ---------------- base-density net PE_3 -> 16 (SiLU) -> 4: first layer on split-fp16 MFMAs with all four products of the
            //                  two-way splits (BSDFD_T32_BASE_SPLIT; 0: seven exact-fp32 MFMAs), second layer fp32 VALU ------------
            {
                const float* Lbw1 = reinterpret_cast<const float*>(smem + LY::BW1);
                const f32x4* Lbb1 = reinterpret_cast<const f32x4*>(smem + LY::BB1 + h * 32);
                const f32x4 b0 = Lbb1[0], b1 = Lbb1[1];
                f32x16 bz = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3], 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const float be[7] = {yh, e[0], e[1], e[2], e[3], e[4], e[5]};
                if (BSDFD_T32_BASE_SPLIT) {
                    const float b03[4] = {be[0], be[1], be[2], be[3]}, b47[4] = {be[4], be[5], be[6], 0.0f};
                    Frag xh, xl;
                    split_pack<true>(b03, xh.p[0], xh.p[1], xl.p[0], xl.p[1]);
                    split_pack<true>(b47, xh.p[2], xh.p[3], xl.p[2], xl.p[3]);
                    const f16x8* Ls = reinterpret_cast<const f16x8*>(smem + LY::BW1S) + lane;
                    const f16x8 ah = Ls[0], al = Ls[64];
                    bz = mfma32(ah, xh.v, bz); bz = mfma32(ah, xl.v, bz); bz = mfma32(al, xh.v, bz); bz = mfma32(al, xl.v, bz);
                } else {
#pragma unroll
                    for (int j = 0; j < 7; ++j) bz = mfma32f(Lbw1[j * 64 + lane], be[j], bz);
                }
                // (the wait states behind the last MFMA are spelled out: see the note at the base net in bsdfd.hip)
                asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(bz));
                const f32x4* Lbw2 = reinterpret_cast<const f32x4*>(smem + LY::BW2 + h * 128);
                f32x4 po = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int v = 0; v < 8; ++v) po += silu(bz[v]) * Lbw2[v];
                // sum over the query's two lanes; every lane needs all four outputs
                float o0 = po[0], o1 = po[1], o2 = po[2], o3 = po[3];
                swap32(o0, o1);
                swap32(o2, o3);
                float s01 = o0 + o1, s23 = o2 + o3;   // lower: out0, out2 | upper: out1, out3
                float a0, a1, a2, a3;
                both32(s01, a0, a1);
                both32(s23, a2, a3);
                const f32x4 b2 = *reinterpret_cast<const f32x4*>(smem + LY::BB2);
                bo = (f32x4){a0 + b2[0], a1 + b2[1], a2 + b2[2], a3 + b2[3]};
            }
            if (!FUSED && p.ctx_out != nullptr) {
                f32x4* c = reinterpret_cast<f32x4*>(p.ctx_out) + ctx_slot * CTX_V4;
#pragma unroll
                for (int k = 0; k < 4; ++k) c[k * 64 + lane] = (f32x4){cacc[4 * k], cacc[4 * k + 1], cacc[4 * k + 2], cacc[4 * k + 3]};
                if (h == 0) c[4 * 64 + n] = bo;
            }
        }
        if constexpr (CaccLds<DOMAIN, FUSED>::on) {
#pragma unroll
            for (int k = 0; k < 4; ++k) cslab[k * 64] = (f32x4){cacc[4 * k], cacc[4 * k + 1], cacc[4 * k + 2], cacc[4 * k + 3]};
        }
        // bo = (loc0, loc1, ls0, ls1) disk | (loc, log_scale, mu, kappa_raw) spherical
        float kappa = 0.0f;
        if (SPH) kappa = softplus(bo[3]) + 1e-3f;

        int ph = 0;
    next_phase:
        {
        const int op = FUSED ? (ph ? OP_PDF : OP_SAMPLE) : p.op;
        const bool reverse = FUSED ? (ph != 0) : reverse1;
        const float cstep = FUSED ? (ph ? -invT : invT) : cstep1;
        float* const out_pdf = (FUSED && ph) ? p.out_pdf2 : p.out_pdf;
        if (FUSED && ph) {
            bool v2;
            load_dir(p.in_c, user_row(row_of(opaque(n), v2)));
        }
        if (FUSED && !ph) { xs = h ? xi1 : xi0; xo0 = xi0; xo1 = xi1; }
        // ---------------- initial state ------------------------------------------------------------
        auto fexp = [](float x) -> float { return __builtin_amdgcn_exp2f(x * kLog2e); };
        if (op == OP_SAMPLE && p.in_b == nullptr) {  // draw x0 ~ D_base(. | omega_i): same counters and arithmetic as bsdfd.hip
            const unsigned long long ctr = p.offset + (unsigned long long)(p.rng_index ? p.rng_index[qi] : qu);
            const unsigned k0 = (unsigned)p.seed, k1 = (unsigned)(p.seed >> 32);
            unsigned u[4];
            philox4x32(k0, k1, (unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0x476175u, u);  // "Gau"
            const float rad = __builtin_amdgcn_sqrtf(-2.0f * kLn2 * __builtin_amdgcn_logf(u01_open(u[0])));
            const float rev = u01_open(u[1]);
            const float sn = __builtin_amdgcn_sinf(rev), cs = __builtin_amdgcn_cosf(rev);
            if (!SPH) {  // model.py:387-392
                xo0 = bo[0] + rad * cs * fexp(bo[2]);
                xo1 = bo[1] + rad * sn * fexp(bo[3]);
            } else {     // model.py:298-307
                xo0 = bo[0] + rad * cs * (fexp(bo[1]) + 1e-3f);
                xo1 = von_mises_sample32(bo[2], kappa, k0, k1, (unsigned)ctr, (unsigned)(ctr >> 32), lane);
            }
            xs = h ? xo1 : xo0;
        }
        auto base_pdf = [&](float a0, float a1) -> float {
            const float log2pi = 1.8378770664093453f;
            if (!SPH) {  // model.py:393-398
                const float e0 = (a0 - bo[0]) * fexp(-bo[2]);
                const float e1 = (a1 - bo[1]) * fexp(-bo[3]);
                return fexp(-log2pi - (bo[2] + bo[3]) - 0.5f * (e0 * e0 + e1 * e1));
            } else {     // model.py:308-317
                const float e = (a0 - bo[0]) * __builtin_amdgcn_rcpf(fexp(bo[1]) + 1e-3f);
                const float loggau = -0.5f * log2pi - bo[1] - 0.5f * e * e;
                float sd_, cd_;
                sincos_enc(a1 - bo[2], sd_, cd_);
                const float logvon = kappa * cd_ - log2pi - log_i0(kappa);
                return fexp(loggau + logvon);
            }
        };
        float p0 = 1.0f;
        if (op == OP_SAMPLE) p0 = base_pdf(xo0, xo1);

        // ---------------- T explicit Euler steps ---------------------------------------------------
        float acc = 1.0f;
        for (int t = 0; t < p.T; ++t) {
            asm volatile("s_nop 0");   // keeps the weight-fragment loads inside the loop (bsdfd.hip, the same statement)
            float alpha;
            if (t_pow2) {
                const float tf = (float)t * invT;
                alpha = reverse ? 1.0f - tf : tf;
            } else {
                const double tf = (double)t * invT_d;
                alpha = (float)(reverse ? 1.0 - tf : tf);
            }
            // ---- layer 1: one fp16 MFMA, B slots [v_hi, v_lo, v_hi, v_lo, w_hi, w_lo, w_hi, w_lo] against A slots
            //      [Wv_hi, Wv_hi, Wv_lo, Wv_lo, Ww_hi, Ww_hi, Ww_lo, Ww_lo]: all four products of the two-way splits ----
            float vin = xs, win = alpha, sp = 0.f, cp = 0.f;
            if (SPH) {
                sincos_enc(xs, sp, cp);   // (meaningful in the upper half, whose xs is phi)
                vin = h ? sp : xs;
                win = h ? cp : alpha;
            }
            const float vh = (BSDFD_SPLIT_RN || BSDFD_T32_L1_RN) ? hi_part_rn(vin) : hi_part(vin), wh = (BSDFD_SPLIT_RN || BSDFD_T32_L1_RN) ? hi_part_rn(win) : hi_part(win);
            const f16x2 zero2 = {(_Float16)0.0f, (_Float16)0.0f};
            const f16x2 vp = {(_Float16)vh, (_Float16)(vin - vh)}, wp = {(_Float16)wh, (_Float16)(win - wh)};
            Frag b1;
            b1.p[0] = vp; b1.p[1] = vp; b1.p[2] = wp; b1.p[3] = wp;
            const f16x8 a1 = *(reinterpret_cast<const f16x8*>(smem + LY::A1) + lane);
            f32x16 cin = cacc;
            if constexpr (CaccLds<DOMAIN, FUSED>::on) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 t4 = cslab[k * 64];
#pragma unroll
                    for (int r = 0; r < 4; ++r) cin[4 * k + r] = t4[r];
                }
            }
            f32x16 z = mfma32(a1, b1.v, cin);

            float hv[16], gv[16];
            Frag bh[2], bl[2], gh[2], gl[2];
            f32x16 U0, U1;  // JAC: gm . U_i, U_i = the tangents' pre-activations at the middle layer, gm = its silu' (fp32)
            float gm[16];
            auto premul = [&] {   // ... multiplied in as soon as gm exists: 16 registers less across the last layer
#pragma unroll
                for (int v = 0; v < 16; ++v) { U0[v] *= gm[v]; U1[v] *= gm[v]; }
            };
            if (!SPH) {
                // ---- MIM, disk 25-32-32-32-2: U_i = F_i g1, R_j = G_j g3, J_ji = sum_k R_j[k] g2[k] U_i[k] (bsdfd.hip, block MIM)
                if constexpr (SPLIT) { act16<JAC>(z, hv, gv); split16(hv, bh, bl); }
                else act_pack16(z, bh);
                z = mm6<SPLIT>(load_mat(smem, LY::WH, lane), bh, bl);
                if constexpr (JAC) {
                    split16_jac(gv, gh, gl);
                    mm6x2(load_mat(smem, LY::WF, lane), gh, gl, load_mat(smem, LY::WF + 4 * FR32, lane), gh, gl, U0, U1);
                }
                // hidden layer 2 (its silu' stays in fp32)
                if constexpr (SPLIT) { act16<JAC>(z, hv, gm); if constexpr (JAC) premul(); split16(hv, bh, bl); }
                else act_pack16(z, bh);
                z = mm6<SPLIT>(load_mat(smem, LY::WH + 4 * FR32, lane), bh, bl);
            } else {
                // ---- MIMS, spherical 26-32-32-32-32-2: two forward-mode tangent layers, then the output fold (bsdfd.hip, block MIMS)
                f32x16 zt0, zt1;
                if constexpr (JAC) {
                    // d(input)/dphi = (0, cos phi, -sin phi, 0): the upper lanes' slots [c_hi, c_lo, c_hi, c_lo, -s_hi, -s_lo, -s_hi, -s_lo]
                    // against the same A fragment; the lower lanes (theta, alpha) contribute nothing
                    Frag bt;
                    bt.p[0] = h ? wp : zero2; bt.p[1] = bt.p[0];
                    bt.p[2] = h ? -vp : zero2; bt.p[3] = bt.p[2];
                    zt1 = mfma32(a1, bt.v, zero16);
                    if (!BSDFD_T32_SPH_FOLD_T0) {
                        const f32x4* Lwt0 = reinterpret_cast<const f32x4*>(smem + LY::WT0) + lane * 4;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const f32x4 t4 = Lwt0[k];
#pragma unroll
                            for (int r = 0; r < 4; ++r) zt0[4 * k + r] = t4[r];
                        }
                    }
                }
#pragma unroll
                for (int layer = 0; layer < 2; ++layer) {
                    if constexpr (SPLIT) { act16<JAC>(z, hv, gv); split16(hv, bh, bl); }
                    else act_pack16(z, bh);
                    const Mat32 w = load_mat(smem, LY::WH + layer * 4 * FR32, lane);
                    z = mm6<SPLIT>(w, bh, bl);
                    if constexpr (JAC) {
                        float t0v[16], t1v[16];
                        const bool fold = BSDFD_T32_SPH_FOLD_T0 && layer == 0;   // W2 (g1 . W1[:, theta]) = F_theta g1
#pragma unroll
                        for (int v = 0; v < 16; ++v) {
                            t0v[v] = fold ? gv[v] : zt0[v] * gv[v];
                            t1v[v] = zt1[v] * gv[v];
                        }
                        Frag t0h[2], t0l[2], t1h[2], t1l[2];
                        split16_jac(t0v, t0h, t0l);
                        split16_jac(t1v, t1h, t1l);
                        if (fold) mm6x2(load_mat(smem, LY::WF, lane), t0h, t0l, w, t1h, t1l, zt0, zt1);
                        else mm6x2(w, t0h, t0l, w, t1h, t1l, zt0, zt1);
                    }
                }
                if constexpr (JAC) { U0 = zt0; U1 = zt1; }
                // hidden layer 3 (its silu' stays in fp32)
                if constexpr (SPLIT) { act16<JAC>(z, hv, gm); if constexpr (JAC) premul(); split16(hv, bh, bl); }
                else act_pack16(z, bh);
                z = mm6<SPLIT>(load_mat(smem, LY::WH + 2 * 4 * FR32, lane), bh, bl);
            }
            // ---- last hidden layer -> R0, R1 (MFMA), v (fp32 VALU dot over the lane's 16 units + one swap) ----
            act16<JAC>(z, hv, gv);
            f32x16 R0, R1;
            if constexpr (JAC) {
                split16_jac(gv, gh, gl);
                mm6x2(load_mat(smem, LY::WG, lane), gh, gl, load_mat(smem, LY::WG + 4 * FR32, lane), gh, gl, R0, R1);
            }
            {
                const f32x4* Lwo = reinterpret_cast<const f32x4*>(smem + LY::WOUT + h * 128);
                f32x2 pv = {0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const f32x4 w = Lwo[k];
                    pv = __builtin_elementwise_fma((f32x2){hv[2 * k], hv[2 * k]}, (f32x2){w[0], w[1]}, pv);
                    pv = __builtin_elementwise_fma((f32x2){hv[2 * k + 1], hv[2 * k + 1]}, (f32x2){w[2], w[3]}, pv);
                }
                float pv0 = pv[0], pv1 = pv[1];
                swap32(pv0, pv1);            // lower: v0 over the lane pair | upper: v1
                xs = fmaf(cstep, pv0 + pv1, xs);
            }
            // ---- J_ji = sum_k R_j[k] (gm[k] U_i[k]) over the lane's 16 units, then over the query's 2 lanes; det(I + c J) ----
            if constexpr (JAC) {
                f32x2 ja2 = {0.f, 0.f}, jb2 = ja2, jc2 = ja2, jd2 = ja2;  // J00, J11, J01, J10
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const f32x2 r0 = {R0[2 * k], R0[2 * k + 1]}, r1 = {R1[2 * k], R1[2 * k + 1]};
                    const f32x2 u0 = {U0[2 * k], U0[2 * k + 1]}, u1 = {U1[2 * k], U1[2 * k + 1]};   // (gm . U_i)
                    ja2 = __builtin_elementwise_fma(r0, u0, ja2);
                    jc2 = __builtin_elementwise_fma(r0, u1, jc2);
                    jd2 = __builtin_elementwise_fma(r1, u0, jd2);
                    jb2 = __builtin_elementwise_fma(r1, u1, jb2);
                }
                float ja = ja2[0] + ja2[1], jb = jb2[0] + jb2[1], jc = jc2[0] + jc2[1], jd = jd2[0] + jd2[1];
                swap32(ja, jb);
                const float sab = ja + jb;    // lower: J00 | upper: J11
                swap32(jc, jd);
                const float scd = jc + jd;    // lower: J01 | upper: J10
                float w = fmaf(cstep, sab, 1.0f), o = cstep * scd;
                swap32(w, o);                 // lower: (w00, w11) | upper: (o01, o10)
                float pr = w * o, pr2 = pr;
                swap32(pr, pr2);              // lower: pr = w00 w11, pr2 = o01 o10
                const float det = pr - pr2;   // valid in the lower half (the lanes that write the results)
                // forward: the reference divides (tmp_J /= J); v_rcp_f32 (1 ulp) * acc differs from the IEEE quotient by <= 2 ulp
                if (reverse) acc *= det; else acc *= __builtin_amdgcn_rcpf(det);
            }
        }

        // ---------------- epilogue: density, warp, guards, store ------------------------------------
        float x0, x1;
        both32(xs, x0, x1);
        float pdf = 0.0f;
        if (op == OP_SAMPLE) pdf = p0 * acc;
        else if (op == OP_PDF) pdf = base_pdf(x0, x1) * acc;

        bool valid_e;
        const long long qe_tile = row_of(opaque(n), valid_e);
        const long long qe = p.io == IO_OPERATOR ? qe_tile : user_row(qe_tile);
        const bool writer = valid_e && h == 0;
        if (p.io == IO_OPERATOR) {
            if (writer) {
                if (op != OP_PDF) reinterpret_cast<float2*>(p.out_x)[qe] = make_float2(x0, x1);
                if (op != OP_SAMPLES_ONLY) out_pdf[qe] = pdf;
            }
        } else if (op == OP_SAMPLE) {
            float ox, oy, oz, pdf_sa;
            if (!SPH) {  // rendering/brdf_measured_disk.py:69-82
                const float r2 = x0 * x0 + x1 * x1;
                const bool ok = r2 < 0.995f;
                ox = ok ? x0 : 0.0f; oy = ok ? x1 : 0.0f;
                oz = sqrtf(fmaxf(1.0f - (ox * ox + oy * oy), 0.0f));
                pdf_sa = (ok ? pdf : 0.0f) * oz;
            } else {  // rendering/brdf_measured_spherical.py:79-91, bsdf_myresult.py:69-84
                float st, ct, sp, cp;
                sincos_enc(x0, st, ct);
                sincos_enc(x1, sp, cp);
                if (!(st > 0.00005f)) pdf = 0.0f;
                if (p.io == IO_PLUGIN && !(ct > 0.0f)) pdf = 0.0f;
                ox = cp * st; oy = sp * st; oz = ct;
                const float inv = fminf(fmaxf(1.0f / sqrtf(ox * ox + oy * oy), 1.0f), 3.402823466e+38f);
                pdf_sa = pdf * inv;
            }
            if (writer) {
                p.out_x[qe * 3 + 0] = ox; p.out_x[qe * 3 + 1] = oy; p.out_x[qe * 3 + 2] = oz;
                out_pdf[qe] = pdf_sa;
            }
        } else {
            float pdf_sa;
            if (!SPH) {  // rendering/brdf_measured_disk.py:112-124
                pdf_sa = (wi_z > 0.0f && wo_z > 0.0f) ? pdf * wo_z : 0.0f;
            } else {
                const float inv = fminf(fmaxf(1.0f / wo_sin, 1.0f), 3.402823466e+38f);
                if (p.io == IO_PLUGIN) {  // rendering/brdf_measured_spherical.py:122-137
                    if (wo_pole) pdf = 0.0f;
                    pdf_sa = (wi_z > 0.0f && wo_z > 0.0f) ? pdf * inv : 0.0f;
                } else {                  // rendering/bsdf_myresult.py:115-133
                    pdf_sa = pdf * inv;
                }
            }
            if (writer) out_pdf[qe] = pdf_sa;
        }
        }
        if (FUSED && ++ph < nphase) goto next_phase;
    }
    if (p.clk) {
        const unsigned long long dc = (unsigned long long)__builtin_readcyclecounter() - clk_c0;
        const unsigned long long dr = (unsigned long long)__builtin_amdgcn_s_memrealtime() - clk_r0;
        if ((threadIdx.x & 63) == 0) {
            unsigned long long* slot = p.clk + 8 * ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (CLK_SLOTS - 1));
            atomicAdd(slot, dc);
            atomicAdd(slot + 1, dr);
        }
    }
}

inline uint16_t f16_bits(float x) {
    const _Float16 hh = (_Float16)x;
    uint16_t b;
    std::memcpy(&b, &hh, 2);
    return b;
}
inline float f16_rnd(float x) { return (float)(_Float16)x; }

template <int DOMAIN>
std::vector<char> build_image32_t(const bsdfd_desc& d) {
    using LY = L32<DOMAIN>;
    constexpr bool SPH = LY::SPH;
    constexpr int NH = LY::NH, W = 32;
    const int SD = SPH ? 3 : 2, IN = SD + 1 + 2 + 4 * PE_BANDS, BIN = 2 + 4 * BASE_PE_BANDS;
    // scaled pre-activation convention (flow_dev.h: silu_grad_scaled): first layer x -log2(e), output layer x -ln 2
    std::vector<float> w_in((size_t)W * IN), w_out((size_t)2 * W);
    for (size_t i = 0; i < w_in.size(); ++i) w_in[i] = (float)((double)d.w_in[i] * -1.4426950408889634);
    for (size_t i = 0; i < w_out.size(); ++i) w_out[i] = (float)((double)d.w_out[i] * -0.6931471805599453);
    std::vector<char> img(LY::TOTAL, 0);
    auto F = [&](int o) { return reinterpret_cast<float*>(img.data() + o); };
    auto H = [&](int o) { return reinterpret_cast<uint16_t*>(img.data() + o); };
    // one matrix = 4 fragments (hi c0, hi c1, lo c0, lo c1); val(row, k) in double
    auto put_matrix = [&](int off, auto&& val) {
        for (int c = 0; c < 2; ++c)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int hh = l >> 5, row = l & 31, k = unit32(8 * c + j, hh);
                    const double v = val(row, k);
                    const float hi = f16_rnd((float)v);
                    H(off)[(size_t)(c * 64 + l) * 8 + j] = f16_bits((float)v);
                    H(off)[(size_t)((2 + c) * 64 + l) * 8 + j] = f16_bits((float)(v - (double)hi));
                }
    };
    // layer-1 state fragment: slots [Wv_hi, Wv_hi, Wv_lo, Wv_lo, Ww_hi, Ww_hi, Ww_lo, Ww_lo] (and the same columns as fp32 A operands)
    for (int l = 0; l < 64; ++l) {
        const int hh = l >> 5, row = l & 31;
        float wv, ww;
        if (!SPH) {   // columns [x0, x1, alpha]
            wv = w_in[(size_t)row * IN + hh];
            ww = hh == 0 ? w_in[(size_t)row * IN + 2] : 0.0f;
        } else {      // columns [theta, sin phi, cos phi, alpha]
            wv = w_in[(size_t)row * IN + (hh == 0 ? 0 : 1)];
            ww = w_in[(size_t)row * IN + (hh == 0 ? 3 : 2)];
        }
        const float s[8] = {f16_rnd(wv), f16_rnd(wv), wv - f16_rnd(wv), wv - f16_rnd(wv), f16_rnd(ww), f16_rnd(ww), ww - f16_rnd(ww), ww - f16_rnd(ww)};
        for (int j = 0; j < 8; ++j) H(LY::A1)[(size_t)l * 8 + j] = f16_bits(s[j]);

    }
    if (SPH && !BSDFD_T32_SPH_FOLD_T0)
        for (int l = 0; l < 64; ++l)
            for (int v = 0; v < 16; ++v) F(LY::WT0)[(size_t)l * 16 + v] = w_in[(size_t)unit32(v, l >> 5) * IN + 0];
    // conditioning term: encoded value index ei (= 2 band + fn; 10 = the raw coordinate) of dimension hh
    auto pe_col = [&](int ei, int hh) { return ei < 10 ? SD + 1 + 2 + 4 * (ei >> 1) + 2 * (ei & 1) + hh : SD + 1 + hh; };
    if (!SPH || BSDFD_T32_SPH_COND_SPLIT) {
        for (int c = 0; c < 2; ++c)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int hh = l >> 5, row = l & 31, ei = 8 * c + j;
                    const float w = ei < 11 ? w_in[(size_t)row * IN + pe_col(ei, hh)] : 0.0f;
                    H(LY::WC)[(size_t)(c * 64 + l) * 8 + j] = f16_bits(w);
                    H(LY::WC)[(size_t)((2 + c) * 64 + l) * 8 + j] = f16_bits(w - f16_rnd(w));
                }
    } else {
        for (int j = 0; j < 11; ++j)
            for (int l = 0; l < 64; ++l) F(LY::WC)[(size_t)j * 64 + l] = w_in[(size_t)(l & 31) * IN + pe_col(j, l >> 5)];
    }
    for (int layer = 0; layer < NH - 1; ++layer)
        put_matrix(LY::WH + layer * 4 * FR32, [&](int row, int k) { return (double)d.w_hidden[((size_t)layer * W + row) * W + k]; });
    if (!SPH)
        for (int i = 0; i < 2; ++i)   // F_i = W2 diag(W1[:, i]) (scaled layer-1 column)
            put_matrix(LY::WF + i * 4 * FR32,
                       [&](int row, int k) { return (double)d.w_hidden[(size_t)row * W + k] * (double)w_in[(size_t)k * IN + i]; });
    else if (BSDFD_T32_SPH_FOLD_T0)   // F_theta = W2 diag(W1[:, theta])
        put_matrix(LY::WF, [&](int row, int k) { return (double)d.w_hidden[(size_t)row * W + k] * (double)w_in[(size_t)k * IN + 0]; });
    for (int i = 0; i < 2; ++i)       // G_i[unit][k] = W_NH[k][unit] Wout[i][k] (Wout scaled by -ln 2)
        put_matrix(LY::WG + i * 4 * FR32, [&](int row, int k) {
            return (double)d.w_hidden[((size_t)(NH - 2) * W + k) * W + row] * (double)w_out[(size_t)i * W + k];
        });
    for (int hh = 0; hh < 2; ++hh)
        for (int v = 0; v < 16; ++v) {
            F(LY::WOUT)[hh * 32 + 2 * v] = w_out[unit32(v, hh)];
            F(LY::WOUT)[hh * 32 + 2 * v + 1] = w_out[W + unit32(v, hh)];
        }
    // base net: inputs [y0, y1, PE_3]; PE entry (band b, fn f, dim dd) at column 2 + 4 b + 2 f + dd
    for (int j = 0; j < 7; ++j)
        for (int l = 0; l < 64; ++l) {
            const int hh = l >> 5, row = l & 31;
            const int col = j == 0 ? hh : 2 + 4 * ((j - 1) >> 1) + 2 * ((j - 1) & 1) + hh;
            F(LY::BW1)[(size_t)j * 64 + l] = row < BASE_HIDDEN ? d.base_w1[(size_t)row * BIN + col] : 0.0f;
        }
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
            const int hh = l >> 5, row = l & 31;
            const int col = j == 0 ? hh : 2 + 4 * ((j - 1) >> 1) + 2 * ((j - 1) & 1) + hh;
            const float w = (j < 7 && row < BASE_HIDDEN) ? d.base_w1[(size_t)row * BIN + col] : 0.0f;
            H(LY::BW1S)[(size_t)l * 8 + j] = f16_bits(w);
            H(LY::BW1S)[(size_t)(64 + l) * 8 + j] = f16_bits(w - f16_rnd(w));
        }
    for (int hh = 0; hh < 2; ++hh)
        for (int v = 0; v < 8; ++v) {
            F(LY::BB1)[hh * 8 + v] = d.base_b1[unit32(v, hh)];
            for (int j = 0; j < 4; ++j) F(LY::BW2)[(hh * 8 + v) * 4 + j] = d.base_w2[(size_t)j * BASE_HIDDEN + unit32(v, hh)];
        }
    for (int j = 0; j < 4; ++j) F(LY::BB2)[j] = d.base_b2[j];
    return img;
}

// ---------------------------------------------------------------------------------------------
// The reflow TEACHER on 32-query tiles: the reference's 64 x 6 spherical net (NN_cond_pos_spherical_complicate,
// rendering/utils/model.py:449-477), single fp16 product, no Jacobian — bsdfd_flow_samples_only in precision f16, the library's
// stand-in for the reference's only tiny-cuda-nn call site (learning_repo_cleanup/spherical_domain_sampling.py:147-166).
// 64 units = two M-tiles of the 32x32 shape: lane (h, n) holds, for query n, units 32 mt + u(v, h) in z[mt][v]; a contraction is
// 2 M-tiles x 4 K-chunks = 8 MFMAs (chunk c = 2 mt' + c': registers 8 c' .. 8 c' + 7 of z[mt'], rounded to fp16, ARE its B
// fragment).  Layer 1: one MFMA per M-tile with the state as hi + lo ([v_hi, v_lo, w_hi, w_lo] against [Wv, Wv, Ww, Ww]: the state
// itself is not rounded to 11 bits, the weights are, as in every other layer of this mode); the conditioning term exact fp32 once
// per query; the output layer an fp32 VALU dot.  Operator io only (the teacher has no plugin form).
// ---------------------------------------------------------------------------------------------
struct L32W {
    static constexpr int NH = 6;
    static constexpr int A1 = 0;                              // 2 fragments (M-tile 0, 1)
    static constexpr int WC = A1 + 2 * FR32;                  // conditioning: 2 M-tiles x 11 x 64 floats
    static constexpr int WH = WC + 2 * 11 * 256;              // (NH - 1) matrices x 2 M-tiles x 4 K-chunks fragments
    static constexpr int WOUT = WH + (NH - 1) * 8 * FR32;     // 2 halves x 2 M-tiles x 16 x (Wout[0][u], Wout[1][u]) floats
    static constexpr int TOTAL = WOUT + 2 * 2 * 128;
};

// 512 threads per workgroup: 8 waves share one 48-KiB image, 2 workgroups per CU = 4 waves per SIMD (116 VGPRs).
__global__ __launch_bounds__(512, 4) void flow_kernel32w(const KParams p) {
    using LY = L32W;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (p.clk) { clk_c0 = __builtin_readcyclecounter(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.img);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (int i = threadIdx.x; i < LY::TOTAL / 16; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const int lane0 = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves_per_block = blockDim.x >> 6;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto uniform_f = [](float x) -> float { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x))); };
    auto uniform_d = [](double x) -> double {
        const unsigned long long b = (unsigned long long)__double_as_longlong(x);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    const double invT_d = uniform_d(1.0 / (double)p.T);
    const bool t_pow2 = (p.T & (p.T - 1)) == 0;
    const float invT = uniform_f((float)invT_d);
    const long long ntiles = (p.N + 31) / 32;
    const int cl = p.chunk_log2;
    const long long chunk = (long long)waves_per_block << cl;
    for (long long it = 0;; ++it) {
        const long long chunk_base = ((it >> cl) * gridDim.x + blockIdx.x) * chunk;
        if (chunk_base >= ntiles) break;
        const long long tile = chunk_base + (it & ((1 << cl) - 1)) * waves_per_block + wave;
        if (tile >= ntiles) continue;
        auto opaque = [](int x) -> int { asm volatile("" : "+v"(x)); return x; };
        const int lane = opaque(lane0);
        const int h = lane >> 5, n = lane & 31;
        const long long qi_raw = tile * 32 + n;
        const long long qi = qi_raw < p.N ? qi_raw : p.N - 1;
        const float2 c2 = reinterpret_cast<const float2*>(p.in_a)[qi];
        const float2 b2 = reinterpret_cast<const float2*>(p.in_b)[qi];
        const float yh = h ? c2.y : c2.x;
        float xs = h ? b2.y : b2.x;
        // conditioning term, exact fp32 (rendering/utils/model.py:26-57 encoding, W1[:, PE] PE(omega_i))
        f32x16 cacc[2] = {zero16, zero16};
        {
            float e[11];
#pragma unroll
            for (int b = 0; b < PE_BANDS; ++b) sincos_enc(yh * (float)(1 << b), e[2 * b], e[2 * b + 1]);
            e[10] = yh;
            const float* Lwc = reinterpret_cast<const float*>(smem + LY::WC);
#pragma unroll
            for (int j = 0; j < 11; ++j) {
                cacc[0] = mfma32f(Lwc[j * 64 + lane], e[j], cacc[0]);
                cacc[1] = mfma32f(Lwc[(11 + j) * 64 + lane], e[j], cacc[1]);
            }
        }
        for (int t = 0; t < p.T; ++t) {
            asm volatile("s_nop 0");   // keeps the weight-fragment loads inside the loop
            float alpha;
            if (t_pow2) alpha = (float)t * invT;
            else alpha = (float)((double)t * invT_d);
            float sp, cp;
            sincos_enc(xs, sp, cp);
            const float vin = h ? sp : xs, win = h ? cp : alpha;
            const float vh = hi_part(vin), wh = hi_part(win);
            Frag b1;
            b1.p[0] = (f16x2){(_Float16)vh, (_Float16)(vin - vh)};
            b1.p[1] = (f16x2){(_Float16)wh, (_Float16)(win - wh)};
            b1.p[2] = b1.p[3] = (f16x2){(_Float16)0.0f, (_Float16)0.0f};
            const f16x8* La1 = reinterpret_cast<const f16x8*>(smem + LY::A1) + lane;
            f32x16 z[2];
            z[0] = mfma32(La1[0], b1.v, cacc[0]);
            z[1] = mfma32(La1[64], b1.v, cacc[1]);
            float hv[2][16];
#pragma unroll
            for (int layer = 0; layer < LY::NH; ++layer) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int v = 0; v < 16; ++v)   // the last layer feeds the fp32 output dot: fp32 sigmoids
                        if (layer == LY::NH - 1) hv[mt][v] = z[mt][v] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z[mt][v]));
                if (layer == LY::NH - 1) break;
                Frag fr[4];   // every other layer: packed-fp16 sigmoids straight into the B fragments (act_pack8)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float z8[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) z8[k] = z[c >> 1][8 * (c & 1) + k];
                    act_pack8(z8, fr[c]);
                }
                const f16x8* W = reinterpret_cast<const f16x8*>(smem + LY::WH + layer * 8 * FR32) + lane;
#pragma unroll
                for (int mo = 0; mo < 2; ++mo) {
                    f32x16 a = mfma32(W[(mo * 4 + 0) * 64], fr[0].v, zero16);
                    a = mfma32(W[(mo * 4 + 1) * 64], fr[1].v, a);
                    a = mfma32(W[(mo * 4 + 2) * 64], fr[2].v, a);
                    z[mo] = mfma32(W[(mo * 4 + 3) * 64], fr[3].v, a);
                }
            }
            const f32x4* Lwo = reinterpret_cast<const f32x4*>(smem + LY::WOUT + h * 256);
            f32x2 pv = {0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const f32x4 w = Lwo[mt * 8 + k];
                    pv = __builtin_elementwise_fma((f32x2){hv[mt][2 * k], hv[mt][2 * k]}, (f32x2){w[0], w[1]}, pv);
                    pv = __builtin_elementwise_fma((f32x2){hv[mt][2 * k + 1], hv[mt][2 * k + 1]}, (f32x2){w[2], w[3]}, pv);
                }
            float pv0 = pv[0], pv1 = pv[1];
            swap32(pv0, pv1);
            xs = fmaf(invT, pv0 + pv1, xs);
        }
        float x0, x1;
        both32(xs, x0, x1);
        const int ne = opaque(n);
        const long long qe = tile * 32 + ne;
        if (qe < p.N && (opaque(lane0) >> 5) == 0) reinterpret_cast<float2*>(p.out_x)[qe] = make_float2(x0, x1);
    }
    if (p.clk) {
        const unsigned long long dc = (unsigned long long)__builtin_readcyclecounter() - clk_c0;
        const unsigned long long dr = (unsigned long long)__builtin_amdgcn_s_memrealtime() - clk_r0;
        if ((threadIdx.x & 63) == 0) {
            unsigned long long* slot = p.clk + 8 * ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (CLK_SLOTS - 1));
            atomicAdd(slot, dc);
            atomicAdd(slot + 1, dr);
        }
    }
}

std::vector<char> build_image32w(const bsdfd_desc& d) {
    using LY = L32W;
    constexpr int W = 64, NH = LY::NH, SD = 3, IN = SD + 1 + 2 + 4 * PE_BANDS;
    std::vector<float> w_in((size_t)W * IN), w_out((size_t)2 * W);
    for (size_t i = 0; i < w_in.size(); ++i) w_in[i] = (float)((double)d.w_in[i] * -1.4426950408889634);
    for (size_t i = 0; i < w_out.size(); ++i) w_out[i] = (float)((double)d.w_out[i] * -0.6931471805599453);
    std::vector<char> img(LY::TOTAL, 0);
    auto F = [&](int o) { return reinterpret_cast<float*>(img.data() + o); };
    auto H = [&](int o) { return reinterpret_cast<uint16_t*>(img.data() + o); };
    auto pe_col = [&](int ei, int hh) { return ei < 10 ? SD + 1 + 2 + 4 * (ei >> 1) + 2 * (ei & 1) + hh : SD + 1 + hh; };
    for (int mt = 0; mt < 2; ++mt)
        for (int l = 0; l < 64; ++l) {
            const int hh = l >> 5, row = 32 * mt + (l & 31);
            const float wv = w_in[(size_t)row * IN + (hh == 0 ? 0 : 1)], ww = w_in[(size_t)row * IN + (hh == 0 ? 3 : 2)];
            const float sl[8] = {wv, wv, ww, ww, 0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < 8; ++j) H(LY::A1)[(size_t)(mt * 64 + l) * 8 + j] = f16_bits(sl[j]);
            for (int j = 0; j < 11; ++j) F(LY::WC)[(size_t)(mt * 11 + j) * 64 + l] = w_in[(size_t)row * IN + pe_col(j, hh)];
        }
    for (int layer = 0; layer < NH - 1; ++layer)
        for (int mo = 0; mo < 2; ++mo)
            for (int c = 0; c < 4; ++c)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int hh = l >> 5, row = 32 * mo + (l & 31), k = 32 * (c >> 1) + unit32(8 * (c & 1) + j, hh);
                        H(LY::WH)[((size_t)(layer * 8 + mo * 4 + c) * 64 + l) * 8 + j] = f16_bits(d.w_hidden[((size_t)layer * W + row) * W + k]);
                    }
    for (int hh = 0; hh < 2; ++hh)
        for (int mt = 0; mt < 2; ++mt)
            for (int v = 0; v < 16; ++v) {
                const int u = 32 * mt + unit32(v, hh);
                F(LY::WOUT)[hh * 64 + mt * 32 + 2 * v] = w_out[u];
                F(LY::WOUT)[hh * 64 + mt * 32 + 2 * v + 1] = w_out[W + u];
            }
    return img;
}

// ---------------------------------------------------------------------------------------------
// The 64 x 6 spherical net WITH the Jacobian on 32-query tiles (round 6): NN_cond_pos_spherical_complicate,
// rendering/utils/model.py:449-477 — the "64-wide" variant north_star asks the kernel to cover — in split-fp16 arithmetic, every call
// that tracks the determinant (network_sampling / network_pdf / plugin sample / plugin pdf; mode 1).  Until round 5 this net ran on
// 16-query tiles only (csrc/bsdfd.hip, FOLDOUT: 227 VGPRs, 0.146 of peak).
//   * 64 units = two M-tiles of the 32x32 shape (as flow_kernel32w): lane (h, n) holds, for query n, units 32 mt + u(v, h) in z[mt][v];
//     a 64 x 64 matrix = 2 x 2 blocks of 32 x 32, each 4 fragments (hi c0, hi c1, lo c0, lo c1) exactly like the 32-wide kernels'.
//   * the Jacobian runs in plain FORWARD mode through all six layers, vector by vector: the activation, then d/dtheta, then d/dphi go
//     through a layer's 16 fragments (12 MFMAs per block pair: 3 products x 2 K-chunks x 2 output M-tiles, alternating the two
//     accumulators).  No folded matrices: the 16-query kernel's output fold buys it 1-2 % and costs 32 KB of LDS, which here holds
//     the conditioning term instead;
//   * d/dtheta of the layer-1 pre-activation is one more MFMA against the state fragment (B slots [1, 0, 1, 0, 0 ...] in the lower
//     half-wave: W_hi + W_lo), d/dphi as in flow_kernel32; the output layer and its two tangent rows are fp32 VALU dots;
//   * the conditioning term (32 registers per lane) lives in a per-wave LDS slab (8 KiB x 8 waves) next to the 92-KiB weight image:
//     512 threads per workgroup, one workgroup per CU, 2 waves per SIMD.
// Same io / segment / context / rng_index / row_index semantics as flow_kernel32 (spherical branch); base draws bit-identical.
// ---------------------------------------------------------------------------------------------
struct L32C {
    static constexpr int NH = 6;
    static constexpr int A1 = 0;                              // layer-1 state fragments of M-tile 0, 1
    static constexpr int WC = A1 + 2 * FR32;                  // conditioning: 2 M-tiles x 11 x 64 floats (exact fp32 A operands)
    static constexpr int WH = WC + 2 * 11 * 256;              // (NH - 1) matrices x blocks (mo, mi) x 4 fragments
    static constexpr int WOUT = WH + (NH - 1) * 16 * FR32;    // 2 halves x 2 M-tiles x 16 x (Wout[0][u], Wout[1][u]) floats
    static constexpr int BW1 = WOUT + 512;                    // base net, as L32
    static constexpr int BB1 = BW1 + 7 * 256;
    static constexpr int BW2 = BB1 + 64;
    static constexpr int BB2 = BW2 + 256;
    static constexpr int BW1S = BB2 + 16;
    static constexpr int TOTAL = BW1S + 2 * FR32;
    static constexpr int SLAB = 64 * 128;                     // per wave: the conditioning accumulators of both M-tiles
    static constexpr int WAVES = 8;
};

__global__ __launch_bounds__(512, 2) void flow_kernel32c(const KParams p) {
    using LY = L32C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (p.clk) { clk_c0 = __builtin_readcyclecounter(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    const char* img = p.img;
    long long q_begin = 0, q_end = p.N;
    int blk = blockIdx.x, nblk = gridDim.x, cl = p.chunk_log2;
    int sidx = 0;
    if (p.nseg > 0) {
        for (int i = 1; i < p.nseg; ++i)
            if ((int)blockIdx.x >= p.seg[i].blk_begin) sidx = i;
        img = p.seg[sidx].img;
        q_begin = p.seg[sidx].q_begin;
        q_end = p.seg[sidx].q_end;
        blk = blockIdx.x - p.seg[sidx].blk_begin;
        nblk = p.seg[sidx].blk_end - p.seg[sidx].blk_begin;
        cl = p.seg[sidx].chunk_log2;
    }
    {
        const uint4* src = reinterpret_cast<const uint4*>(img);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (int i = threadIdx.x; i < LY::TOTAL / 16; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();

    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: the tile index and everything derived from it stay in SGPRs)
    const int waves_per_block = blockDim.x >> 6;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto uniform_f = [](float x) -> float { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x))); };
    auto uniform_d = [](double x) -> double {
        const unsigned long long b = (unsigned long long)__double_as_longlong(x);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    const double invT_d = uniform_d(1.0 / (double)p.T);
    const bool t_pow2 = (p.T & (p.T - 1)) == 0;
    const float invT = uniform_f((float)invT_d);
    const bool reverse = (p.op == OP_PDF);
    const float cstep = reverse ? -invT : invT;
    const int op = p.op;
    const long long ntiles = (q_end - q_begin + 31) / 32;
    const long long chunk = (long long)waves_per_block << cl;
    for (long long it = 0;; ++it) {
        const long long chunk_base = ((it >> cl) * nblk + blk) * chunk;
        if (chunk_base >= ntiles) break;
        const long long tile = chunk_base + (it & ((1 << cl) - 1)) * waves_per_block + wave;
        if (tile >= ntiles) continue;
        auto opaque = [](int x) -> int { asm volatile("" : "+v"(x)); return x; };
        const int lane = opaque(lane0);
        const int h = lane >> 5;
        const int n = lane & 31;
        const long long tile_q0 = q_begin + tile * 32;
        auto row_of = [&](int nn, bool& in_range) -> long long {
            const long long r = tile_q0 + nn;
            in_range = r < q_end;
            return in_range ? r : q_end - 1;
        };
        bool in_range0;
        const long long qi = row_of(n, in_range0);
        auto user_row = [&](long long r) -> long long { return p.row_index ? p.row_index[r] : r; };
        const long long qu = user_row(qi);

        // ---------------- inputs (flow_kernel32, spherical branch) -------------------------------------
        // (what the pdf epilogue needs of wi / wo — cos theta_i, cos theta_o, sin theta_o, the pole flag — is re-read there instead of
        //  being carried across the Euler loop: the loop has no register to spare at 2 waves per SIMD)
        float yh = 0.f;
        float xs = 0.f;
        float xo0 = 0.f, xo1 = 0.f;
        const bool have_ctx = p.ctx_in != nullptr;
        if (p.io == IO_OPERATOR) {
            const float2 c2 = reinterpret_cast<const float2*>(p.in_a)[qi];
            yh = h ? c2.y : c2.x;
            if (p.op == OP_PDF || p.in_b != nullptr) {
                const float2 b2 = reinterpret_cast<const float2*>(p.in_b)[qi];
                xs = h ? b2.y : b2.x;
                xo0 = b2.x; xo1 = b2.y;
            }
        } else {
            const float wx = p.in_a[qu * 3 + 0], wy = p.in_a[qu * 3 + 1], wz = p.in_a[qu * 3 + 2];
            if (!have_ctx) {
                const SphArgs ai = spher_args(wx, wy, wz);  // rendering/brdf_measured_spherical.py:35-39
                yh = atan2f(h ? ai.y : ai.s, h ? ai.x : ai.z);
            }
            if (p.op == OP_PDF) {
                const float ox = p.in_b[qu * 3 + 0], oy = p.in_b[qu * 3 + 1], oz = p.in_b[qu * 3 + 2];
                const SphArgs ao = spher_args(ox, oy, oz);
                xs = atan2f(h ? ao.y : ao.s, h ? ao.x : ao.z);
            } else if (p.in_b != nullptr) {  // injected base sample
                const float2 b2 = reinterpret_cast<const float2*>(p.in_b)[qu];
                xs = h ? b2.y : b2.x; xo0 = b2.x; xo1 = b2.y;
            }
        }

        f32x16 cacc[2];
        f32x4* const cslab = reinterpret_cast<f32x4*>(smem + LY::TOTAL + wave * LY::SLAB) + lane;   // [k][lane] f32x4, k = 4 mt + 0..3
        f32x4 bo;
        constexpr long long CTX_V4 = 8 * 64 + 32;  // f32x4 per tile: cacc (8 per lane) + bo per query
        const long long ctx_slot = ((q_begin + tile * 32) >> 5) + p.seg_base + sidx;
        if (have_ctx) {
            const f32x4* c = reinterpret_cast<const f32x4*>(p.ctx_in) + ctx_slot * CTX_V4;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const f32x4 t4 = c[k * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) cacc[k >> 2][4 * (k & 3) + r] = t4[r];
            }
            bo = c[8 * 64 + n];
        } else {
            float e[11];
#pragma unroll
            for (int b = 0; b < PE_BANDS; ++b) sincos_enc(yh * (float)(1 << b), e[2 * b], e[2 * b + 1]);
            e[10] = yh;
            // conditioning term c = W1[:, PE] PE(omega_i): exact fp32 chains, both M-tiles
            const float* Lwc = reinterpret_cast<const float*>(smem + LY::WC);
            cacc[0] = zero16; cacc[1] = zero16;
#pragma unroll
            for (int j = 0; j < 11; ++j) {
                cacc[0] = mfma32f(Lwc[j * 64 + lane], e[j], cacc[0]);
                cacc[1] = mfma32f(Lwc[(11 + j) * 64 + lane], e[j], cacc[1]);
            }
            // base-density net PE_3 -> 16 (SiLU) -> 4: as flow_kernel32
            {
                const f32x4* Lbb1 = reinterpret_cast<const f32x4*>(smem + LY::BB1 + h * 32);
                const f32x4 b0 = Lbb1[0], b1 = Lbb1[1];
                f32x16 bz = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3], 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const float b03[4] = {yh, e[0], e[1], e[2]}, b47[4] = {e[3], e[4], e[5], 0.0f};
                Frag xh, xl;
                split_pack<true>(b03, xh.p[0], xh.p[1], xl.p[0], xl.p[1]);
                split_pack<true>(b47, xh.p[2], xh.p[3], xl.p[2], xl.p[3]);
                const f16x8* Ls = reinterpret_cast<const f16x8*>(smem + LY::BW1S) + lane;
                const f16x8 ah = Ls[0], al = Ls[64];
                bz = mfma32(ah, xh.v, bz); bz = mfma32(ah, xl.v, bz); bz = mfma32(al, xh.v, bz); bz = mfma32(al, xl.v, bz);
                asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(bz));
                const f32x4* Lbw2 = reinterpret_cast<const f32x4*>(smem + LY::BW2 + h * 128);
                f32x4 po = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int v = 0; v < 8; ++v) po += silu(bz[v]) * Lbw2[v];
                float o0 = po[0], o1 = po[1], o2 = po[2], o3 = po[3];
                swap32(o0, o1);
                swap32(o2, o3);
                float s01 = o0 + o1, s23 = o2 + o3;
                float a0, a1, a2, a3;
                both32(s01, a0, a1);
                both32(s23, a2, a3);
                const f32x4 b2 = *reinterpret_cast<const f32x4*>(smem + LY::BB2);
                bo = (f32x4){a0 + b2[0], a1 + b2[1], a2 + b2[2], a3 + b2[3]};
            }
            if (p.ctx_out != nullptr) {
                f32x4* c = reinterpret_cast<f32x4*>(p.ctx_out) + ctx_slot * CTX_V4;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    c[k * 64 + lane] = (f32x4){cacc[k >> 2][4 * (k & 3)], cacc[k >> 2][4 * (k & 3) + 1], cacc[k >> 2][4 * (k & 3) + 2], cacc[k >> 2][4 * (k & 3) + 3]};
                if (h == 0) c[8 * 64 + n] = bo;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            cslab[k * 64] = (f32x4){cacc[k >> 2][4 * (k & 3)], cacc[k >> 2][4 * (k & 3) + 1], cacc[k >> 2][4 * (k & 3) + 2], cacc[k >> 2][4 * (k & 3) + 3]};
        auto kappa_of = [&]() -> float { return softplus(bo[3]) + 1e-3f; };   // (recomputed where needed, not carried)

        // ---------------- initial state ------------------------------------------------------------
        auto fexp = [](float x) -> float { return __builtin_amdgcn_exp2f(x * kLog2e); };
        if (op == OP_SAMPLE && p.in_b == nullptr) {  // draw x0 ~ D_base(. | omega_i): same counters and arithmetic as the other kernels
            const unsigned long long ctr = p.offset + (unsigned long long)(p.rng_index ? p.rng_index[qi] : qu);
            const unsigned k0 = (unsigned)p.seed, k1 = (unsigned)(p.seed >> 32);
            unsigned u[4];
            philox4x32(k0, k1, (unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0x476175u, u);  // "Gau"
            const float rad = __builtin_amdgcn_sqrtf(-2.0f * kLn2 * __builtin_amdgcn_logf(u01_open(u[0])));
            const float rev = u01_open(u[1]);
            const float cs = __builtin_amdgcn_cosf(rev);
            xo0 = bo[0] + rad * cs * (fexp(bo[1]) + 1e-3f);   // model.py:298-307
            xo1 = von_mises_sample32(bo[2], kappa_of(), k0, k1, (unsigned)ctr, (unsigned)(ctr >> 32), lane);
            xs = h ? xo1 : xo0;
        }
        auto base_pdf = [&](float a0, float a1) -> float {   // model.py:308-317
            const float log2pi = 1.8378770664093453f;
            const float e = (a0 - bo[0]) * __builtin_amdgcn_rcpf(fexp(bo[1]) + 1e-3f);
            const float loggau = -0.5f * log2pi - bo[1] - 0.5f * e * e;
            float sd_, cd_;
            sincos_enc(a1 - bo[2], sd_, cd_);
            const float kappa = kappa_of();
            const float logvon = kappa * cd_ - log2pi - log_i0(kappa);
            return fexp(loggau + logvon);
        };
        // ---------------- T explicit Euler steps ---------------------------------------------------
        float acc = 1.0f;   // sampling: starts at p0 and is divided by every det — (p0 / det_1) / det_2 ... as the reference's tmp_J /= J
        if (op == OP_SAMPLE) acc = base_pdf(xo0, xo1);
        for (int t = 0; t < p.T; ++t) {
            asm volatile("s_nop 0");   // keeps the weight-fragment loads inside the loop
            float alpha;
            if (t_pow2) {
                const float tf = (float)t * invT;
                alpha = reverse ? 1.0f - tf : tf;
            } else {
                const double tf = (double)t * invT_d;
                alpha = (float)(reverse ? 1.0 - tf : tf);
            }
            float sp, cp;
            sincos_enc(xs, sp, cp);   // (meaningful in the upper half, whose xs is phi)
            const float vin = h ? sp : xs, win = h ? cp : alpha;
            const float vh = hi_part(vin), wh = hi_part(win);
            const f16x2 zero2 = {(_Float16)0.0f, (_Float16)0.0f}, one2 = {(_Float16)1.0f, (_Float16)0.0f};
            const f16x2 vp = {(_Float16)vh, (_Float16)(vin - vh)}, wp = {(_Float16)wh, (_Float16)(win - wh)};
            Frag b1, bt, bt0;
            b1.p[0] = vp; b1.p[1] = vp; b1.p[2] = wp; b1.p[3] = wp;
            // d(input)/dphi = (0, cos phi, -sin phi, 0) in the upper lanes; d(input)/dtheta = (1, 0, 0, 0) in the lower ones
            bt.p[0] = h ? wp : zero2; bt.p[1] = bt.p[0]; bt.p[2] = h ? -vp : zero2; bt.p[3] = bt.p[2];
            bt0.p[0] = h ? zero2 : one2; bt0.p[1] = bt0.p[0]; bt0.p[2] = zero2; bt0.p[3] = zero2;
            const f16x8* La1 = reinterpret_cast<const f16x8*>(smem + LY::A1) + lane;
            f32x16 z[2], zt0[2], zt1[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x16 cin;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 t4 = cslab[(4 * mt + k) * 64];
#pragma unroll
                    for (int r = 0; r < 4; ++r) cin[4 * k + r] = t4[r];
                }
                const f16x8 a1 = La1[mt * 64];
                z[mt] = mfma32(a1, b1.v, cin);
                zt0[mt] = mfma32(a1, bt0.v, zero16);
                zt1[mt] = mfma32(a1, bt.v, zero16);
            }
            // ---- hidden layers 1 .. NH-1: activation + the two tangents through W2 .. W_NH ----
            // the three vectors (activation, d/dtheta, d/dphi) share every weight fragment: one 32 x 32 block at a time, 18 MFMAs
            // round-robin over the three accumulators
            // (the vector-by-vector order — each vector re-reading the 16 fragments — needs fewer registers and measured 5 % slower still)
#pragma unroll 1
            for (int layer = 0; layer < LY::NH - 1; ++layer) {
                Frag hh[2][2], hl[2][2], ah[2][2], al[2][2], bh[2][2], bl[2][2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    float hv[16], gv[16], t0v[16], t1v[16];
                    act16<true>(z[mt], hv, gv);
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        t0v[v] = zt0[mt][v] * gv[v];
                        t1v[v] = zt1[mt][v] * gv[v];
                    }
                    split16(hv, hh[mt], hl[mt]);
                    split16(t0v, ah[mt], al[mt]);
                    split16(t1v, bh[mt], bl[mt]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mo = 0; mo < 2; ++mo) {
                    f32x16 a = zero16, b = zero16, c = zero16;
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
                        const Mat32 w = load_mat(smem, LY::WH + ((layer * 2 + mo) * 2 + mi) * 4 * FR32, lane);
                        a = mfma32(w.h0, hh[mi][0].v, a); b = mfma32(w.h0, ah[mi][0].v, b); c = mfma32(w.h0, bh[mi][0].v, c);
                        a = mfma32(w.h1, hh[mi][1].v, a); b = mfma32(w.h1, ah[mi][1].v, b); c = mfma32(w.h1, bh[mi][1].v, c);
                        a = mfma32(w.h0, hl[mi][0].v, a); b = mfma32(w.h0, al[mi][0].v, b); c = mfma32(w.h0, bl[mi][0].v, c);
                        a = mfma32(w.h1, hl[mi][1].v, a); b = mfma32(w.h1, al[mi][1].v, b); c = mfma32(w.h1, bl[mi][1].v, c);
                        a = mfma32(w.l0, hh[mi][0].v, a); b = mfma32(w.l0, ah[mi][0].v, b); c = mfma32(w.l0, bh[mi][0].v, c);
                        a = mfma32(w.l1, hh[mi][1].v, a); b = mfma32(w.l1, ah[mi][1].v, b); c = mfma32(w.l1, bh[mi][1].v, c);
                    }
                    z[mo] = a; zt0[mo] = b; zt1[mo] = c;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- last hidden layer -> v and its two tangent rows: fp32 VALU dots over the lane's 32 units, then the lane pair ----
            {
                const f32x4* Lwo = reinterpret_cast<const f32x4*>(smem + LY::WOUT + h * 256);
                f32x2 pv = {0.f, 0.f}, pd0 = pv, pd1 = pv;   // (v0, v1), d/dtheta of them, d/dphi of them
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    float hv[16], gv[16];
                    act16<true>(z[mt], hv, gv);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const f32x4 w = Lwo[mt * 8 + k];
                        const f32x2 w0 = {w[0], w[1]}, w1 = {w[2], w[3]};
                        const float ta = zt0[mt][2 * k] * gv[2 * k], tb = zt0[mt][2 * k + 1] * gv[2 * k + 1];
                        const float ua = zt1[mt][2 * k] * gv[2 * k], ub = zt1[mt][2 * k + 1] * gv[2 * k + 1];
                        pv = __builtin_elementwise_fma((f32x2){hv[2 * k], hv[2 * k]}, w0, pv);
                        pv = __builtin_elementwise_fma((f32x2){hv[2 * k + 1], hv[2 * k + 1]}, w1, pv);
                        pd0 = __builtin_elementwise_fma((f32x2){ta, ta}, w0, pd0);
                        pd0 = __builtin_elementwise_fma((f32x2){tb, tb}, w1, pd0);
                        pd1 = __builtin_elementwise_fma((f32x2){ua, ua}, w0, pd1);
                        pd1 = __builtin_elementwise_fma((f32x2){ub, ub}, w1, pd1);
                    }
                }
                float pv0 = pv[0], pv1 = pv[1];
                swap32(pv0, pv1);            // lower: v0 over the lane pair | upper: v1
                xs = fmaf(cstep, pv0 + pv1, xs);
                // J00 = dv0/dtheta, J11 = dv1/dphi, J01 = dv0/dphi, J10 = dv1/dtheta; det(I + c J) as in flow_kernel32
                float ja = pd0[0], jb = pd1[1], jc = pd1[0], jd = pd0[1];
                swap32(ja, jb);
                const float sab = ja + jb;    // lower: J00 | upper: J11
                swap32(jc, jd);
                const float scd = jc + jd;    // lower: J01 | upper: J10
                float w = fmaf(cstep, sab, 1.0f), o = cstep * scd;
                swap32(w, o);                 // lower: (w00, w11) | upper: (o01, o10)
                float pr = w * o, pr2 = pr;
                swap32(pr, pr2);              // lower: pr = w00 w11, pr2 = o01 o10
                const float det = pr - pr2;   // valid in the lower half (the lanes that write the results)
                if (reverse) acc *= det; else acc *= __builtin_amdgcn_rcpf(det);
            }
        }

        // ---------------- epilogue (flow_kernel32, spherical branch) ------------------------------------
        float x0, x1;
        both32(xs, x0, x1);
        float pdf = 0.0f;
        if (op == OP_SAMPLE) pdf = acc;
        else if (op == OP_PDF) pdf = base_pdf(x0, x1) * acc;
        bool valid_e;
        const long long qe_tile = row_of(opaque(n), valid_e);
        const long long qe = p.io == IO_OPERATOR ? qe_tile : user_row(qe_tile);
        const bool writer = valid_e && (opaque(lane0) >> 5) == 0;
        if (p.io == IO_OPERATOR) {
            if (writer) {
                if (op != OP_PDF) reinterpret_cast<float2*>(p.out_x)[qe] = make_float2(x0, x1);
                p.out_pdf[qe] = pdf;
            }
        } else if (op == OP_SAMPLE) {  // rendering/brdf_measured_spherical.py:79-91, bsdf_myresult.py:69-84
            float st, ct, sp, cp;
            sincos_enc(x0, st, ct);
            sincos_enc(x1, sp, cp);
            if (!(st > 0.00005f)) pdf = 0.0f;
            if (p.io == IO_PLUGIN && !(ct > 0.0f)) pdf = 0.0f;
            const float ox = cp * st, oy = sp * st, oz = ct;
            const float inv = fminf(fmaxf(1.0f / sqrtf(ox * ox + oy * oy), 1.0f), 3.402823466e+38f);
            if (writer) {
                p.out_x[qe * 3 + 0] = ox; p.out_x[qe * 3 + 1] = oy; p.out_x[qe * 3 + 2] = oz;
                p.out_pdf[qe] = pdf * inv;
            }
        } else {
            const float wi_z = p.in_a[qe * 3 + 2];
            const float ox = p.in_b[qe * 3 + 0], oy = p.in_b[qe * 3 + 1], wo_z = p.in_b[qe * 3 + 2];
            const float wo_sin = sqrtf(ox * ox + oy * oy);  // Mitsuba Frame3f::sin_theta
            const bool wo_pole = spher_args(ox, oy, wo_z).ref_pole;
            const float inv = fminf(fmaxf(1.0f / wo_sin, 1.0f), 3.402823466e+38f);
            float pdf_sa;
            if (p.io == IO_PLUGIN) {  // rendering/brdf_measured_spherical.py:122-137
                if (wo_pole) pdf = 0.0f;
                pdf_sa = (wi_z > 0.0f && wo_z > 0.0f) ? pdf * inv : 0.0f;
            } else {                  // rendering/bsdf_myresult.py:115-133
                pdf_sa = pdf * inv;
            }
            if (writer) p.out_pdf[qe] = pdf_sa;
        }
    }
    if (p.clk) {
        const unsigned long long dc = (unsigned long long)__builtin_readcyclecounter() - clk_c0;
        const unsigned long long dr = (unsigned long long)__builtin_amdgcn_s_memrealtime() - clk_r0;
        if ((threadIdx.x & 63) == 0) {
            unsigned long long* slot = p.clk + 8 * ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (CLK_SLOTS - 1));
            atomicAdd(slot, dc);
            atomicAdd(slot + 1, dr);
        }
    }
}

std::vector<char> build_image32c(const bsdfd_desc& d) {
    using LY = L32C;
    constexpr int W = 64, NH = LY::NH, SD = 3, IN = SD + 1 + 2 + 4 * PE_BANDS, BIN = 2 + 4 * BASE_PE_BANDS;
    std::vector<float> w_in((size_t)W * IN), w_out((size_t)2 * W);
    for (size_t i = 0; i < w_in.size(); ++i) w_in[i] = (float)((double)d.w_in[i] * -1.4426950408889634);
    for (size_t i = 0; i < w_out.size(); ++i) w_out[i] = (float)((double)d.w_out[i] * -0.6931471805599453);
    std::vector<char> img(LY::TOTAL, 0);
    auto F = [&](int o) { return reinterpret_cast<float*>(img.data() + o); };
    auto H = [&](int o) { return reinterpret_cast<uint16_t*>(img.data() + o); };
    auto pe_col = [&](int ei, int hh) { return ei < 10 ? SD + 1 + 2 + 4 * (ei >> 1) + 2 * (ei & 1) + hh : SD + 1 + hh; };
    for (int mt = 0; mt < 2; ++mt)
        for (int l = 0; l < 64; ++l) {
            const int hh = l >> 5, row = 32 * mt + (l & 31);
            // layer-1 state fragment: slots [Wv_hi, Wv_hi, Wv_lo, Wv_lo, Ww_hi, Ww_hi, Ww_lo, Ww_lo]; columns [theta, sin phi, cos phi, alpha]
            const float wv = w_in[(size_t)row * IN + (hh == 0 ? 0 : 1)], ww = w_in[(size_t)row * IN + (hh == 0 ? 3 : 2)];
            const float sl[8] = {f16_rnd(wv), f16_rnd(wv), wv - f16_rnd(wv), wv - f16_rnd(wv), f16_rnd(ww), f16_rnd(ww), ww - f16_rnd(ww), ww - f16_rnd(ww)};
            for (int j = 0; j < 8; ++j) H(LY::A1)[(size_t)(mt * 64 + l) * 8 + j] = f16_bits(sl[j]);
            for (int j = 0; j < 11; ++j) F(LY::WC)[(size_t)(mt * 11 + j) * 64 + l] = w_in[(size_t)row * IN + pe_col(j, hh)];
        }
    // hidden matrices: block (mo, mi) = rows 32 mo .., K units 32 mi + u(8 c + j, hh): 4 fragments (hi c0, hi c1, lo c0, lo c1)
    for (int layer = 0; layer < NH - 1; ++layer)
        for (int mo = 0; mo < 2; ++mo)
            for (int mi = 0; mi < 2; ++mi) {
                const int off = LY::WH + ((layer * 2 + mo) * 2 + mi) * 4 * FR32;
                for (int c = 0; c < 2; ++c)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int hh = l >> 5, row = 32 * mo + (l & 31), k = 32 * mi + unit32(8 * c + j, hh);
                            const float v = d.w_hidden[((size_t)layer * W + row) * W + k];
                            H(off)[(size_t)(c * 64 + l) * 8 + j] = f16_bits(v);
                            H(off)[(size_t)((2 + c) * 64 + l) * 8 + j] = f16_bits(v - f16_rnd(v));
                        }
            }
    for (int hh = 0; hh < 2; ++hh)
        for (int mt = 0; mt < 2; ++mt)
            for (int v = 0; v < 16; ++v) {
                const int u = 32 * mt + unit32(v, hh);
                F(LY::WOUT)[hh * 64 + mt * 32 + 2 * v] = w_out[u];
                F(LY::WOUT)[hh * 64 + mt * 32 + 2 * v + 1] = w_out[W + u];
            }
    // base net: as build_image32_t
    for (int j = 0; j < 7; ++j)
        for (int l = 0; l < 64; ++l) {
            const int hh = l >> 5, row = l & 31;
            const int col = j == 0 ? hh : 2 + 4 * ((j - 1) >> 1) + 2 * ((j - 1) & 1) + hh;
            F(LY::BW1)[(size_t)j * 64 + l] = row < BASE_HIDDEN ? d.base_w1[(size_t)row * BIN + col] : 0.0f;
        }
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
            const int hh = l >> 5, row = l & 31;
            const int col = j == 0 ? hh : 2 + 4 * ((j - 1) >> 1) + 2 * ((j - 1) & 1) + hh;
            const float w = (j < 7 && row < BASE_HIDDEN) ? d.base_w1[(size_t)row * BIN + col] : 0.0f;
            H(LY::BW1S)[(size_t)l * 8 + j] = f16_bits(w);
            H(LY::BW1S)[(size_t)(64 + l) * 8 + j] = f16_bits(w - f16_rnd(w));
        }
    for (int hh = 0; hh < 2; ++hh)
        for (int v = 0; v < 8; ++v) {
            F(LY::BB1)[hh * 8 + v] = d.base_b1[unit32(v, hh)];
            for (int j = 0; j < 4; ++j) F(LY::BW2)[(hh * 8 + v) * 4 + j] = d.base_w2[(size_t)j * BASE_HIDDEN + unit32(v, hh)];
        }
    for (int j = 0; j < 4; ++j) F(LY::BB2)[j] = d.base_b2[j];
    return img;
}

// which 32-query-tile kernel serves (net, precision, mode): 0 none, 1 flow_kernel32 (split3), 2 flow_kernel32w, 3 flow_kernel32 (f16),
// 4 flow_kernel32c (the 64 x 6 spherical net in split3 with the Jacobian: mode 1 only — its samples-only and fused calls stay on
// 16-query tiles).  Kind 4 is OPT-IN (bsdfd_desc.tile = 32 explicitly; bsdfd_kernel32_opt_in): measured 4 % slower than the 16-query
// kernel — 4 % fewer shader cycles at a 7.5 % lower clock, +2.5 % J/query (profiles/r06_ab/ab32c_64x6_jacobian.txt) — the library's
// default for this net stays the 16-query kernel.
#ifndef BSDFD_T32_WIDE_JAC
#define BSDFD_T32_WIDE_JAC 1   // 0: flow_kernel32c not offered at all (A/B builds)
#endif
int kind32(const bsdfd_desc& d, int prec, int mode) {
    if (BSDFD_T32_WIDE_JAC && prec == BSDFD_PREC_SPLIT3 && d.width == 64 && d.n_hidden == 6 && d.domain == BSDFD_DOMAIN_SPHERICAL && mode == 1)
        return 4;
    if (prec == BSDFD_PREC_SPLIT3 && d.width == 32 &&
        ((d.domain == BSDFD_DOMAIN_DISK && d.n_hidden == 3) || (d.domain == BSDFD_DOMAIN_SPHERICAL && d.n_hidden == 4)))
        return 1;
    if (prec == BSDFD_PREC_F16 && d.width == 64 && d.n_hidden == 6 && d.domain == BSDFD_DOMAIN_SPHERICAL && mode == 0) return 2;
    if (prec == BSDFD_PREC_F16 && d.width == 32 && mode == 0 &&
        ((d.domain == BSDFD_DOMAIN_DISK && d.n_hidden == 3) || (d.domain == BSDFD_DOMAIN_SPHERICAL && d.n_hidden == 4)))
        return 3;   // flow_kernel32 with single fp16 products (samples-only)
    return 0;
}

}  // namespace

bool bsdfd_tile32_supported(const bsdfd_desc& d, int prec) {
    return kind32(d, prec, 0) != 0 || kind32(d, prec, 1) != 0;
}

bool bsdfd_kernel32_opt_in(const bsdfd_desc& d, int prec, int mode) { return kind32(d, prec, mode) == 4; }

int bsdfd_tile32_context_v4(const bsdfd_desc& d, int prec) {   // f32x4 records per 32-query tile of the per-query context (mode 1)
    return kind32(d, prec, 1) == 4 ? 8 * 64 + 32 : 4 * 64 + 32;
}


std::vector<char> bsdfd_build_image32(const bsdfd_desc& d, int prec) {
    if (kind32(d, prec, 1) == 4) return build_image32c(d);
    if (kind32(d, prec, 0) == 2) return build_image32w(d);
    return d.domain == BSDFD_DOMAIN_DISK ? build_image32_t<BSDFD_DOMAIN_DISK>(d) : build_image32_t<BSDFD_DOMAIN_SPHERICAL>(d);
}

int bsdfd_kernel32_lds_bytes(const bsdfd_desc& d, int prec, int mode) {
    if (kind32(d, prec, mode) == 4) return L32C::TOTAL + L32C::WAVES * L32C::SLAB;
    if (kind32(d, prec, mode) == 2) return L32W::TOTAL;
    const int domain = d.domain;   // (kind 3 = mode 0 of the same kernel template: same image, same slab)
    if (mode == 2)
        return domain == BSDFD_DOMAIN_DISK ? L32<BSDFD_DOMAIN_DISK>::TOTAL + 4 * CaccLds<BSDFD_DOMAIN_DISK, true>::slab
                                           : L32<BSDFD_DOMAIN_SPHERICAL>::TOTAL + 4 * CaccLds<BSDFD_DOMAIN_SPHERICAL, true>::slab;
    return domain == BSDFD_DOMAIN_DISK ? L32<BSDFD_DOMAIN_DISK>::TOTAL + 4 * CaccLds<BSDFD_DOMAIN_DISK, false>::slab
                                       : L32<BSDFD_DOMAIN_SPHERICAL>::TOTAL + 4 * CaccLds<BSDFD_DOMAIN_SPHERICAL, false>::slab;
}

int bsdfd_kernel32_threads(const bsdfd_desc& d, int prec, int mode) {
    const int kind = kind32(d, prec, mode);
    return (kind == 2 || kind == 4) ? 512 : 256;
}

const void* bsdfd_kernel32(const bsdfd_desc& d, int prec, int mode) {
    const int kind = kind32(d, prec, mode);
    if (kind == 4) return reinterpret_cast<const void*>(flow_kernel32c);
    if (kind == 2) return reinterpret_cast<const void*>(flow_kernel32w);
    const bool disk = d.domain == BSDFD_DOMAIN_DISK;
    if (kind == 3)
        return disk ? reinterpret_cast<const void*>(flow_kernel32<BSDFD_DOMAIN_DISK, false, false, false>)
                    : reinterpret_cast<const void*>(flow_kernel32<BSDFD_DOMAIN_SPHERICAL, false, false, false>);
    if (kind != 1) return nullptr;
    switch (mode) {
        case 0: return disk ? reinterpret_cast<const void*>(flow_kernel32<BSDFD_DOMAIN_DISK, false, false, true>)
                            : reinterpret_cast<const void*>(flow_kernel32<BSDFD_DOMAIN_SPHERICAL, false, false, true>);
        case 1: return disk ? reinterpret_cast<const void*>(flow_kernel32<BSDFD_DOMAIN_DISK, true, false, true>)
                            : reinterpret_cast<const void*>(flow_kernel32<BSDFD_DOMAIN_SPHERICAL, true, false, true>);
        case 2: return disk ? reinterpret_cast<const void*>(flow_kernel32<BSDFD_DOMAIN_DISK, true, true, true>)
                            : reinterpret_cast<const void*>(flow_kernel32<BSDFD_DOMAIN_SPHERICAL, true, true, true>);
    }
    return nullptr;
}
