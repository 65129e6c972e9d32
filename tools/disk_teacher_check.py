#!/usr/bin/env python3
"""The DISK reflow teacher (32 x 3 `brdf_diffusion_network`, learning_repo_cleanup/disk_domain_sampling.py:93-110) through
bsdfd_flow_samples_only in precision f16 on 16- vs 32-query tiles: error against the fp64 oracle on the golden inputs, ragged
sizes, alternating kernel time.      python tools/disk_teacher_check.py [T] [N]"""
import json
import os
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import GOLDEN
from oracle import bsdf_oracle as O
from bsdf_diffusion_sampling_amd import weights as W
from bsdf_diffusion_sampling_amd.sampler import FlowSampler
T = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 22
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
g = np.load(os.path.join(GOLDEN, "disk_teacher_aniso_miro_7_rgb.npz"))
fw = W.load(W.shipped_path("aniso_miro_7_rgb", "disk", "diffusion"))
wi, x0 = g["wi"], g["x0"]
xo, _ = O.Oracle(fw).flow(x0, wi, T, reverse=False)
smp = {tile: FlowSampler(fw, precision="f16", tile=tile) for tile in (16, 32)}
out = {}
for tile, s in smp.items():
    x = s.flow_samples_only(t(wi), t(x0), T=T).cpu().numpy()
    e = np.abs(x - xo).max(1)
    xr = s.flow_samples_only(t(wi[:333]), t(x0[:333]), T=T).cpu().numpy()
    out[f"t{tile}"] = {"tile_samples_only": s.tile_samples_only, "p50": float(np.percentile(e, 50)), "p99": float(np.percentile(e, 99)),
                       "max": float(e.max()), "ragged_equal": bool(np.array_equal(xr, x[:333]))}
print(json.dumps(out))
gen = torch.Generator(device="cuda").manual_seed(1)
r, a = 0.95 * torch.sqrt(torch.rand(n, device="cuda", generator=gen)), 6.2831853 * torch.rand(n, device="cuda", generator=gen)
cond = torch.stack([r * torch.cos(a), r * torch.sin(a)], 1).contiguous()
xs = (0.3 * torch.randn(n, 2, device="cuda", generator=gen)).contiguous()
res = {16: [], 32: []}
for rnd in range(3):
    for tile in (16, 32):
        s = smp[tile]
        for _ in range(2): s.flow_samples_only(cond, xs, T=T)
        torch.cuda.synchronize(); s.set_profiling(True)
        for _ in range(4): s.flow_samples_only(cond, xs, T=T)
        k, ms = s.profile_read(); mhz = s.profile_clock_mhz(); s.set_profiling(False)
        res[tile].append((ms / k, mhz))
for tile in (16, 32):
    print(tile, [(round(a_, 3), round(b_)) for a_, b_ in res[tile]])
print("ratio", np.median([a_ for a_, _ in res[32]]) / np.median([a_ for a_, _ in res[16]]))
