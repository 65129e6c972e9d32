#!/usr/bin/env python3
"""The "one reciprocal for two / four sigmoids" experiment (profiles/HISTORY.md round 5 #15, profiles/r05_ab/ab32_rcp_pairs.txt;
negative): writes build_ab/flow32_var.hip, a COPY of csrc/flow32.hip whose act16 takes -DRCP_MODE=0 (product: one v_rcp_f32 per
sigmoid), 1 (pairs: r = rcp(a b), 1/a = r b), 2 (quads), 4 (quads behind a v_min_f32 clamp of the scaled pre-activation).

    python3 tools/rcp_pairs_variant.py
    CS=$PWD/bsdf_diffusion_sampling_amd/csrc; FLOW32_SRC=$PWD/build_ab/flow32_var.hip bash tools/ab_build32.sh \
        base "-I $CS -DRCP_MODE=0" pair "-I $CS -DRCP_MODE=1" quad "-I $CS -DRCP_MODE=2" quadclamp "-I $CS -DRCP_MODE=4"
    gpurun -- 'BSDFD_TILE=32 bash tools/ab_run.sh 3 "--only disk8,disk4,sph8" base pair quad quadclamp'
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OLD = '''template <bool WITH_G>
__device__ __forceinline__ void act16(const f32x16& z, float (&hs)[16], float (&g)[16]) {
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        if (WITH_G) silu_grad_scaled(z[v], hs[v], g[v]);
        else hs[v] = z[v] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z[v]));
    }
}
'''
NEW = '''#ifndef RCP_MODE
#define RCP_MODE 0
#endif
template <bool WITH_G>
__device__ __forceinline__ void act16(const f32x16& zin, float (&hs)[16], float (&g)[16]) {
    float s[16];
    float z[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) z[v] = zin[v];
#if RCP_MODE == 0
#pragma unroll
    for (int v = 0; v < 16; ++v) s[v] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z[v]));
#elif RCP_MODE == 1
#pragma unroll
    for (int v = 0; v < 16; v += 2) {
        const float a0 = 1.0f + __builtin_amdgcn_exp2f(z[v]), a1 = 1.0f + __builtin_amdgcn_exp2f(z[v + 1]);
        const float r = __builtin_amdgcn_rcpf(a0 * a1);     // overflows where z[v] + z[v + 1] > 128: NaN rows on chm_orange_rgb_disk
        s[v] = r * a1; s[v + 1] = r * a0;
    }
#else
#pragma unroll
    for (int v = 0; v < 16; v += 4) {
#if RCP_MODE == 4
#pragma unroll
        for (int k = 0; k < 4; ++k) z[v + k] = fminf(z[v + k], 30.0f);
#endif
        const float a0 = 1.0f + __builtin_amdgcn_exp2f(z[v]), a1 = 1.0f + __builtin_amdgcn_exp2f(z[v + 1]);
        const float a2 = 1.0f + __builtin_amdgcn_exp2f(z[v + 2]), a3 = 1.0f + __builtin_amdgcn_exp2f(z[v + 3]);
        const float p01 = a0 * a1, p23 = a2 * a3;
        const float r = __builtin_amdgcn_rcpf(p01 * p23);
        const float r01 = r * p23, r23 = r * p01;
        s[v] = r01 * a1; s[v + 1] = r01 * a0; s[v + 2] = r23 * a3; s[v + 3] = r23 * a2;
    }
#endif
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        hs[v] = z[v] * s[v];
        if (WITH_G) g[v] = fmaf(hs[v], fmaf(s[v], kLn2, -kLn2), s[v]);
    }
}
'''


def main():
    src = open(os.path.join(ROOT, "bsdf_diffusion_sampling_amd", "csrc", "flow32.hip")).read()
    assert OLD in src, "act16 of csrc/flow32.hip has changed: update OLD"
    os.makedirs(os.path.join(ROOT, "build_ab"), exist_ok=True)
    out = os.path.join(ROOT, "build_ab", "flow32_var.hip")
    open(out, "w").write(src.replace(OLD, NEW))
    print("wrote", out)


if __name__ == "__main__":
    main()
