#!/usr/bin/env python3
"""Is a launch's time quantised by workgroup rounds?  Kernel time of the disk sample launch (32-query tiles, T = 8) at batch sizes
around whole multiples of the resident workgroup capacity (256 CUs x 3 workgroups x 4 waves x 2^cl tiles x 32 queries), alternating,
HIP events on the launch stream.   python tools/grid_quant.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from bsdf_diffusion_sampling_amd import weights as W  # noqa: E402
from bsdf_diffusion_sampling_amd.sampler import FlowSampler  # noqa: E402

dev = torch.device("cuda", 0)
s = FlowSampler(W.load(W.shipped_path("aniso_miro_7_rgb", "disk")), tile=32)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
per_round = 256 * 3 * 4 * 4 * 32          # cl = 2 at these sizes: 393 216 queries per round of resident workgroups
sizes = [per_round * 2, 1 << 20, per_round * 3, int(per_round * 3.5), per_round * 4, 1 << 21, per_round * 6]
bufs = {n: (bench.make_wi("disk", n, 1234, dev), torch.empty((n, 3), device=dev), torch.empty((n,), device=dev)) for n in sizes}
res = {n: [] for n in sizes}
for rnd in range(6):
    for n in sizes:
        wi, wo, p = bufs[n]
        for _ in range(3):
            s.plugin_sample(wi, None, T=T, seed=1, out=(wo, p))
        torch.cuda.synchronize()
        s.set_profiling(True)
        for _ in range(10):
            s.plugin_sample(wi, None, T=T, seed=1, out=(wo, p))
        k, ms = s.profile_read()
        mhz = s.profile_clock_mhz()
        s.set_profiling(False)
        if rnd:
            res[n].append((ms / k, mhz))
for n in sizes:
    ms = float(np.median([a for a, _ in res[n]])); mhz = float(np.median([b for _, b in res[n]]))
    print(json.dumps({"queries": n, "rounds_of_resident_workgroups": n / per_round, "kernel_us": ms * 1e3, "in_kernel_mhz": mhz,
                      "ns_per_query": ms * 1e6 / n, "Mcycles": ms * mhz * 1e-3}))
