#!/usr/bin/env python3
"""bsdfd_flow_samples_only in precision f16 (the reflow teachers' call; packed-fp16 sigmoids, csrc/flow_dev.h: act_pack8) on EVERY
shipped weight set against the fp64 oracle: 1024 queries, T = 128 Euler steps, both tilings.  What it is for: the class bound on
nets whose pre-activations saturate fp16's exponential (2^zs = inf from zs = 16: chm_orange reaches zs > 60) — no NaN, no row
outside atol + rtol |x| with 1e-2 each.      python3 tools/f16_sweep.py [--out gpurun_out/f16_sweep.json]"""
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
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--T", type=int, default=128)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(77)
    n, T = a.n, a.T
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    rows = {}
    for dom in ("disk", "spherical"):
        for stem in W.list_shipped(dom):
            fw = W.load(os.path.join(W.DATA_DIR, stem + ".bsdfw"))
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
            xo, _ = orc.flow(x032, cond32, T, reverse=False)
            row = {}
            for tile in (32, 16):
                s = FlowSampler(fw, precision="f16", tile=tile)
                x = s.flow_samples_only(t(cond32), t(x032), T=T).cpu().numpy().astype(np.float64)
                err = np.abs(x - xo)
                row[f"tile{s.tile_samples_only}"] = {"p50": float(np.nanpercentile(err, 50)), "p99": float(np.nanpercentile(err, 99)),
                                                      "max": float(np.nanmax(err)), "nan": int(np.isnan(x).sum()),
                                                      "outside_class": int((err > 1e-2 + 1e-2 * np.abs(xo)).sum())}
                s.close()
            rows[stem] = row
            print(stem, row, flush=True)
    worst = {}
    for tile in ("tile32", "tile16"):
        have = {k: v[tile] for k, v in rows.items() if tile in v}
        if have:
            w99 = max(have, key=lambda k: have[k]["p99"])
            wmax = max(have, key=lambda k: have[k]["max"])
            worst[tile] = {"sets": len(have), "worst_p99": have[w99]["p99"], "worst_p99_set": w99, "worst_max": have[wmax]["max"],
                           "worst_max_set": wmax, "nan": sum(v["nan"] for v in have.values()),
                           "rows_outside_class": sum(v["outside_class"] for v in have.values())}
    print(json.dumps(worst))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump({"n": n, "T": T, "summary": worst, "sets": rows}, open(a.out, "w"), indent=0)


if __name__ == "__main__":
    main()
