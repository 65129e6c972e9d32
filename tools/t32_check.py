#!/usr/bin/env python3
"""32-query-tile kernels (csrc/flow32.hip) against the fp64 oracle and against the 16-query-tile kernels, on the golden cases,
plus an interleaved timing of the two tilings (HIP events on the launch stream).  Prints one JSON document.

    python tools/t32_check.py [--reps 7] [--no-time]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bsdf_diffusion_sampling_amd import weights as W  # noqa: E402
from bsdf_diffusion_sampling_amd.sampler import FlowSampler  # noqa: E402


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def rel(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-30)


def accuracy():
    from conftest import GOLDEN_CASES, load_case
    from oracle import bsdf_oracle as O
    res = {}
    for stem in GOLDEN_CASES:
        g, fw = load_case(stem)
        if fw.width != 32:
            continue
        T = int(g["meta_T"])
        orc = O.Oracle(fw)
        xo, po = orc.network_sampling(g["wi"], g["x0"], T)
        _, acc = orc.flow(g["x0"], g["wi"], T, False)
        ok = (np.abs(acc) > 1e-3) & (np.abs(acc) < 1e3)
        ok = ok & (np.abs(po) > 1e-6 * np.percentile(np.abs(po[ok]), 99))
        pr = orc.network_pdf(xo, g["wi"], T)
        ok2 = ok & (np.abs(pr) > 1e-6 * np.percentile(np.abs(pr), 99))
        out = {}
        for tile in (16, 32):
            s = FlowSampler(fw, precision="split3", tile=tile)
            x, p = s.network_sampling(t(g["wi"]), t(g["x0"]), T=T)
            x, p = x.cpu().numpy().astype(np.float64), p.cpu().numpy().astype(np.float64)
            p2 = s.network_pdf(t(xo), t(g["wi"]), T=T).cpu().numpy().astype(np.float64)
            r, r2 = rel(p, po)[ok], rel(p2, pr)[ok2]
            out[f"t{tile}"] = {"x_err": float(np.abs(x - xo).max()), "p99": float(np.percentile(r, 99)), "max": float(r.max()),
                               "pdf_p99": float(np.percentile(r2, 99)), "pdf_max": float(r2.max()),
                               "nan": int(np.isnan(p).sum() + np.isnan(p2).sum() + np.isnan(x).sum())}
            s.close()
        res[stem] = out
    return res


def kernel_ms(smp, fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        smp.set_profiling(True)
        fn()
        n, ms = smp.profile_read()
        out.append(ms / max(n, 1))
    smp.set_profiling(False)
    return float(np.median(out))


def timing(reps):
    res = {}
    gen = torch.Generator(device="cuda").manual_seed(1)
    for name, stem, n, T in (("disk8", "aniso_miro_7_rgb_disk", 1 << 20, 8), ("disk4", "aniso_miro_7_rgb_disk", 1 << 20, 4),
                             ("sph8", "chm_orange_rgb_spherical", 1 << 22, 8)):
        fw = W.load(os.path.join(W.DATA_DIR, stem + ".bsdfw"))
        wi = torch.randn(n, 3, device="cuda", generator=gen)
        wi[:, 2] = wi[:, 2].abs() + 0.05
        wi = wi / wi.norm(dim=1, keepdim=True)
        smp = {tile: FlowSampler(fw, precision="split3", tile=tile) for tile in (16, 32)}
        wo, _ = smp[16].plugin_sample(wi, None, T=T)
        rows = {16: {"sample": [], "pdf": []}, 32: {"sample": [], "pdf": []}}
        for rnd in range(3):
            for tile in (16, 32):
                s = smp[tile]
                rows[tile]["sample"].append(kernel_ms(s, lambda: s.plugin_sample(wi, None, T=T), reps))
                rows[tile]["pdf"].append(kernel_ms(s, lambda: s.plugin_pdf(wi, wo, T=T), reps))
        res[name] = {f"t{tile}_{op}_ms": float(np.median(v)) for tile in (16, 32) for op, v in rows[tile].items()}
        for op in ("sample", "pdf"):
            res[name][f"ratio_{op}"] = res[name][f"t32_{op}_ms"] / res[name][f"t16_{op}_ms"]
        for s in smp.values():
            s.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--no-time", action="store_true")
    a = ap.parse_args()
    out = {"accuracy": accuracy()}
    if not a.no_time:
        out["timing"] = timing(a.reps)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
