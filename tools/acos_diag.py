#!/usr/bin/env python3
"""Is cart_to_spher the ONLY place where the kernel deviates from the reference's fp32 outputs?  

The kernel evaluates theta = acos(z / (r + 1e-8)) (rendering/brdf_measured_spherical.py:35-39) in the well-conditioned form
atan2(sqrt(x^2 + y^2 + 2 r eps + eps^2), z): closer to the fp64 answer than the reference's own fp32 run, and
therefore FURTHER from that fp32 run than 1e-4 on chm_orange (vs_ref32_p99 2.4e-4 / 1.45e-4 on 2 048 rows).  This tool runs the
16 384-row plugin fixture of that material (tests/golden/chm_orange_rgb_spherical_n16k_plugin.npz, produced by running the
reference) through two builds of the 32-query-tile kernels:

    product   libbsdfd.so as shipped
    acosdiag  build_ab/lib_acosdiag.so = csrc/flow32.hip compiled with -DBSDFD_DIAG_ACOS_AS_WRITTEN (acosf of the fp32 quotient,
              the reference's line as written; tools/ab_build32.sh acosdiag "-DBSDFD_DIAG_ACOS_AS_WRITTEN")

and prints, per call, the p99 of |kernel - reference fp32| / |reference fp32| and of |kernel - fp64| / |fp64| (+ bootstrap 95 %
intervals).  If cart_to_spher is the only deviation, the diagnostic build's distance to the reference's fp32 outputs falls to the
fp32 noise floor (< 1e-4) while its distance to the fp64 answer rises to the reference's own.

    python tools/acos_diag.py [--out gpurun_out/r06/acos_diag.json]     (GPU box; the product build never has the knob)"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker():
    import numpy as np
    import torch
    import parity77 as P
    from bsdf_diffusion_sampling_amd import _lib
    from bsdf_diffusion_sampling_amd.sampler import FlowSampler
    from conftest import GOLDEN, load_case
    from oracle import bsdf_oracle as O
    stem = "chm_orange_rgb_spherical_n16k"
    _, fw = load_case(stem)
    p = np.load(os.path.join(GOLDEN, stem + "_plugin.npz"))
    T = int(p["meta_T"])
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()  # noqa: E731
    s = FlowSampler(fw, precision="split3", tile=32, binding="ctypes")
    rng = np.random.default_rng(0)
    out = {"library": _lib.lib().bsdfd_version().decode(), "lib_path": _lib.LIB_PATH, "rows": int(p["wi3"].shape[0])}

    def figures(got, ref32, ref64):
        got, ref32 = got.astype(np.float64), ref32.astype(np.float64)
        ok = (np.abs(ref64) > 1e-6 * np.percentile(np.abs(ref64), 99)) & (ref32 != 0) & (got != 0)
        e32 = np.abs(got - ref32)[ok] / np.abs(ref32[ok])
        e64 = np.abs(got - ref64)[ok] / np.abs(ref64[ok])
        n64 = np.abs(ref32 - ref64)[ok] / np.abs(ref64[ok])
        r = {"rows": int(ok.sum())}
        for k, e in (("vs_ref32", e32), ("vs_fp64", e64), ("ref32_vs_fp64", n64)):
            p99, lo, hi = P._p99_ci(e, rng)
            r[k + "_p99"], r[k + "_p99_ci"] = p99, [lo, hi]
        return r
    wo, pdf = s.plugin_sample(t(p["wi3"]), t(p["x0"]), T=T)
    out["sample"] = figures(pdf.cpu().numpy(), p["sample_pdf_sa"], p["sample_pdf_sa_f64"])
    out["sample"]["wo_vs_ref32_max"] = float(np.abs(wo.cpu().numpy() - p["sample_wo3"]).max())
    out["sample"]["wo_vs_fp64_max"] = float(np.abs(wo.cpu().numpy() - p["sample_wo3_f64"]).max())
    orc = O.Oracle(fw)
    for wi3, wo3, key in ((p["pdf_wi3"], p["pdf_wo3"], "pdf_sa"), (p["wi3"], p["sample_wo3"], "pdf_sa_of_samples")):
        got = s.plugin_pdf(t(wi3), t(wo3), T=T).cpu().numpy()
        want = O.plugin_pdf_spherical(orc, wi3.astype(np.float64), wo3.astype(np.float64), T=T)
        out[key] = figures(got, p[key], want)
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--worker", action="store_true")
    a = ap.parse_args()
    if a.worker:
        return worker()
    res = {}
    for name, lib in (("product", None), ("acosdiag", os.path.join(ROOT, "build_ab", "lib_acosdiag.so"))):
        env = dict(os.environ)
        if lib:
            if not os.path.exists(lib):
                sys.exit(f"{lib} not found: tools/ab_build32.sh acosdiag \"-DBSDFD_DIAG_ACOS_AS_WRITTEN\"")
            env["BSDFD_LIB_PATH"] = lib
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], env=env, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-2000:])
        res[name] = json.loads(r.stdout.strip().splitlines()[-1])
    for name, r in res.items():
        for k in ("sample", "pdf_sa", "pdf_sa_of_samples"):
            print(f"{name:9s} {k:18s} vs the reference's fp32 run p99 {r[k]['vs_ref32_p99']:.2e}   vs fp64 p99 {r[k]['vs_fp64_p99']:.2e}   "
                  f"(reference fp32 vs fp64: {r[k]['ref32_vs_fp64_p99']:.2e}; {r[k]['rows']} rows)")
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
