#!/usr/bin/env python3
"""Operator-level accuracy of ONE build of libbsdfd.so (selected with $BSDFD_LIB_PATH) on every shipped plugin weight set
(27 disk, 25 spherical, 25 full-sphere bsdf_<i>): network_sampling + network_pdf against the fp64 oracle, 2048 queries each.

    BSDFD_LIB_PATH=build_ab/lib_X.so python tools/acc_sweep.py [--tag X] [--n 2048] [--out gpurun_out/acc_X.json]

Rows counted: |prod det J| in [1e-3, 1e3] and p_ref > 1e-6 x its 99th percentile (SURVEY.md §8(d) error metric).
Prints one summary line; --out keeps the per-set table."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bsdf_diffusion_sampling_amd import weights as W  # noqa: E402
from bsdf_diffusion_sampling_amd.sampler import FlowSampler  # noqa: E402
from oracle import bsdf_oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="lib")
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--precision", default="default")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(2024)
    n = a.n
    rows = {}
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    for dom in ("disk", "spherical"):
        T = 4 if dom == "disk" else 8
        for stem in W.list_shipped(dom):
            if stem.endswith("_complex"):
                continue
            fw = W.load(W.shipped_path(stem[: -len(dom) - 1], dom))
            orc = O.Oracle(fw)
            if dom == "disk":
                r, ang = 0.95 * np.sqrt(rng.random(n)), 2 * np.pi * rng.random(n)
                cond = np.stack([r * np.cos(ang), r * np.sin(ang)], 1)
                x0 = orc.base_sample(cond, rng.standard_normal((n, 2)))
            else:
                hi = 3.0 if stem.startswith("bsdf_") else 1.5
                cond = np.stack([hi * rng.random(n), (2 * rng.random(n) - 1) * np.pi], 1)
                mu, kappa = orc.base_von_mises_params(cond)
                x0 = orc.base_sample(cond, rng.standard_normal(n), phi=rng.vonmises(mu, kappa))
            cond32, x032 = cond.astype(np.float32), x0.astype(np.float32)
            s = FlowSampler(fw, precision=a.precision)
            x, p = s.network_sampling(t(cond32), t(x032), T=T)
            x, p = x.cpu().numpy().astype(np.float64), p.cpu().numpy().astype(np.float64)
            xo, po = orc.network_sampling(cond32, x032, T)
            _, acc = orc.flow(x032, cond32, T, reverse=False)
            ok = (np.abs(acc) > 1e-3) & (np.abs(acc) < 1e3)
            ok &= np.abs(po) > 1e-6 * np.percentile(np.abs(po[ok]), 99)
            rel = np.abs(p - po)[ok] / np.abs(po[ok])
            pp = s.network_pdf(t(x.astype(np.float32)), t(cond32), T=T).cpu().numpy().astype(np.float64)
            ppo = orc.network_pdf(x.astype(np.float32), cond32, T)
            _, accr = orc.flow(x.astype(np.float32), cond32, T, reverse=True)
            okr = (np.abs(accr) > 1e-3) & (np.abs(accr) < 1e3)
            okr &= np.abs(ppo) > 1e-6 * np.percentile(np.abs(ppo[okr]), 99)
            relr = np.abs(pp - ppo)[okr] / np.abs(ppo[okr])
            rows[stem] = {"sample_p99": float(np.percentile(rel, 99)), "sample_med": float(np.median(rel)), "sample_max": float(rel.max()),
                          "pdf_p99": float(np.percentile(relr, 99)), "pdf_max": float(relr.max()),
                          "x_err_max": float(np.abs(x - xo)[ok].max()), "nan": int(np.isnan(p).sum() + np.isnan(pp).sum())}
            s.close()
    worst_s = max(rows.items(), key=lambda kv: kv[1]["sample_p99"])
    worst_p = max(rows.items(), key=lambda kv: kv[1]["pdf_p99"])
    summ = {"tag": a.tag, "sets": len(rows), "queries_per_set": n, "precision": a.precision,
            "worst_sample_p99": [worst_s[0], worst_s[1]["sample_p99"]], "worst_pdf_p99": [worst_p[0], worst_p[1]["pdf_p99"]],
            "median_of_sample_p99": float(np.median([r["sample_p99"] for r in rows.values()])),
            "median_of_pdf_p99": float(np.median([r["pdf_p99"] for r in rows.values()])),
            "worst_x_err": max(r["x_err_max"] for r in rows.values()), "nan": sum(r["nan"] for r in rows.values())}
    if a.out:
        json.dump({"summary": summ, "sets": rows}, open(a.out, "w"), indent=1)
    print(json.dumps(summ), flush=True)


if __name__ == "__main__":
    main()
