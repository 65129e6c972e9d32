#!/usr/bin/env python3
"""Pin the ground-truth evaluator (csrc/measured.hip, oracle/measured_oracle.py) against Mitsuba's own `measured` plugin —
on the day a box with Mitsuba 3 exists.  Nothing in this image can import `mitsuba` (DESIGN.md §7 records the
install attempt), so row f3 stays "parity unpinned"; this script is the missing half, ready to run:

    pip install mitsuba            # any machine with network access
    python tests/golden/make_mitsuba_golden.py

It evaluates what the reference's plugins delegate to (rendering/brdf_measured_disk.py:36-42: `mi.load_dict({'type':
'measured', 'filename': ...})`, `:103-110`: `self.bsdf.eval(ctx, si, wo)`) on the one tensor file shipped as a fixture
(tests/golden/chm_orange_rgb.bsdf) for a fixed grid of direction pairs and writes tests/golden/mitsuba_measured_eval.npz
(wi, wo, f cos as Mitsuba returns it, and its pdf()).  tests/test_measured_cpu.py picks the file up when present and holds
the oracle to it."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    try:
        import mitsuba as mi
    except ImportError as exc:
        print(f"mitsuba is not importable here ({exc}); nothing written.  Run this where `pip install mitsuba` works.")
        return 2
    for variant in ("llvm_ad_rgb", "scalar_rgb"):
        if variant in mi.variants():
            mi.set_variant(variant)
            break
    else:
        print("no rgb variant of mitsuba available:", mi.variants())
        return 2
    bsdf = mi.load_dict({"type": "measured", "filename": os.path.join(HERE, "chm_orange_rgb.bsdf")})
    rng = np.random.default_rng(20251002)
    n = 4096
    def hemi(k):
        z = rng.uniform(0.02, 1.0, size=k)
        ph = rng.uniform(0, 2 * np.pi, size=k)
        r = np.sqrt(1 - z * z)
        return np.stack([r * np.cos(ph), r * np.sin(ph), z], 1).astype(np.float32)
    wi, wo = hemi(n), hemi(n)
    wo[: n // 4] = wi[: n // 4] * np.array([-1, -1, 1], np.float32)          # the mirror direction: the lobe's peak
    wo[n // 4: n // 2] += 0.05 * rng.standard_normal((n // 4, 3)).astype(np.float32)
    wo[n // 4: n // 2, 2] = np.abs(wo[n // 4: n // 2, 2]) + 1e-3
    wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    ctx = mi.BSDFContext()
    f = np.zeros((n, 3), np.float32)
    pdf = np.zeros(n, np.float32)
    if mi.variant().startswith("scalar"):
        for i in range(n):
            si = mi.SurfaceInteraction3f()
            si.wi = mi.Vector3f(*map(float, wi[i]))
            f[i] = np.array(bsdf.eval(ctx, si, mi.Vector3f(*map(float, wo[i]))))
            pdf[i] = float(bsdf.pdf(ctx, si, mi.Vector3f(*map(float, wo[i]))))
    else:
        import drjit as dr
        si = dr.zeros(mi.SurfaceInteraction3f, n)
        si.wi = mi.Vector3f(wi[:, 0], wi[:, 1], wi[:, 2])
        w = mi.Vector3f(wo[:, 0], wo[:, 1], wo[:, 2])
        f = np.array(bsdf.eval(ctx, si, w)).reshape(3, -1).T.astype(np.float32) if np.array(bsdf.eval(ctx, si, w)).shape[0] == 3 \
            else np.array(bsdf.eval(ctx, si, w)).astype(np.float32)
        pdf = np.array(bsdf.pdf(ctx, si, w)).astype(np.float32)
    out = os.path.join(HERE, "mitsuba_measured_eval.npz")
    np.savez_compressed(out, wi=wi, wo=wo, f_cos=f, pdf=pdf, meta_mitsuba=mi.__version__, meta_variant=mi.variant())
    print("wrote", out, f.shape, "max f cos", float(f.max()))
    return 0


if __name__ == "__main__":
    sys.exit(main())
