"""Kernel-time A/B of the per-query context (include/bsdfd.h, bsdfd_context_bytes): sample() with / without writing it,
pdf() with / without reading it.  HIP events on the launch stream, interleaved, median of `reps` launches each.
    python3 tools/ctx_ab.py [reps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from bsdf_diffusion_sampling_amd import weights as W  # noqa: E402
from bsdf_diffusion_sampling_amd.sampler import FlowSampler  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
dev = torch.device("cuda", 0)


def timed(s, fn):
    s.set_profiling(True)
    fn()
    n, ms = s.profile_read()
    s.set_profiling(False)
    return ms / max(n, 1)


for dom, T, N in (("disk", 8, 1 << 20), ("disk", 4, 1 << 20), ("spherical", 8, 1 << 20), ("spherical", 8, 1 << 24)):
    s = FlowSampler(W.load(W.shipped_path("aniso_miro_7_rgb", dom)))
    wi = bench.make_wi(dom, N, 1234, dev)
    ctx = s.new_context(N)
    wo, pdf = s.plugin_sample(wi, None, T=T, seed=1, ctx_out=ctx)
    out_p = torch.empty_like(pdf)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:  # leave the idle clocks
        s.plugin_pdf(wi, wo, T=T, out=out_p)
        torch.cuda.synchronize()
    res = {k: [] for k in ("sample", "sample+ctx_out", "pdf", "pdf+ctx_in")}
    for _ in range(reps):
        res["sample"].append(timed(s, lambda: s.plugin_sample(wi, None, T=T, seed=1, out=(wo, pdf))))
        res["sample+ctx_out"].append(timed(s, lambda: s.plugin_sample(wi, None, T=T, seed=1, out=(wo, pdf), ctx_out=ctx)))
        res["pdf"].append(timed(s, lambda: s.plugin_pdf(wi, wo, T=T, out=out_p)))
        res["pdf+ctx_in"].append(timed(s, lambda: s.plugin_pdf(wi, wo, T=T, out=out_p, ctx_in=ctx)))
    med = {k: float(np.median(v)) for k, v in res.items()}
    pair0, pair1 = med["sample"] + med["pdf"], med["sample+ctx_out"] + med["pdf+ctx_in"]
    print(json.dumps({"domain": dom, "T": T, "N": N, "median_ms": {k: round(v, 4) for k, v in med.items()},
                      "pair_ms": [round(pair0, 4), round(pair1, 4)], "pair_ratio": round(pair1 / pair0, 4),
                      "ctx_MB": round(ctx.numel() * 4 / 1e6, 1)}), flush=True)
