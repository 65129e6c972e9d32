"""Tiny driver for profiler passes that want few launches of ONE kernel: the disk T=8 sample() kernel on 1 Mi queries
(BASELINE.json configs[1]).  Usage: python3 tools/pcsamp_run.py [launches] [workload: disk8|disk4|sph8]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from bsdf_diffusion_sampling_amd import weights as W  # noqa: E402
from bsdf_diffusion_sampling_amd.sampler import FlowSampler  # noqa: E402

n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 10
wl = sys.argv[2] if len(sys.argv) > 2 else "disk8"
dom, T, N = {"disk8": ("disk", 8, 1 << 20), "disk4": ("disk", 4, 1 << 20), "sph8": ("spherical", 8, 1 << 22)}[wl]
dev = torch.device("cuda", 0)
s = FlowSampler(W.load(W.shipped_path("aniso_miro_7_rgb", dom)))
wi = bench.make_wi(dom, N, 1234, dev)
for _ in range(n_launch):
    wo, pdf = s.plugin_sample(wi, None, T=T, seed=1)
    s.plugin_pdf(wi, wo, T=T)
torch.cuda.synchronize()
print("done", wl, n_launch)
