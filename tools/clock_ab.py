#!/usr/bin/env python3
"""Shader clock the chip sustains under the 16- and the 32-query-tile kernels (in-kernel s_memtime / s_memrealtime counters of the
launches themselves), alternating, with the kernel time next to it:  python tools/clock_ab.py [disk|spherical] [T] [N]"""
import sys
sys.path.insert(0, '.')
import time
import numpy as np, torch
import bench
from bsdf_diffusion_sampling_amd import weights as W
from bsdf_diffusion_sampling_amd.sampler import FlowSampler
dom = sys.argv[1] if len(sys.argv) > 1 else "disk"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
dev = torch.device("cuda")
fw = W.load(W.shipped_path("aniso_miro_7_rgb", dom))
smp = {t: FlowSampler(fw, tile=t) for t in (16, 32)}
wi = bench.make_wi(dom, n, 1234, dev)
wo = torch.empty((n, 3), device=dev); p = torch.empty(n, device=dev)
t0 = time.time()
while time.time() - t0 < 0.3:
    smp[16].plugin_sample(wi, None, T=T, out=(wo, p)); torch.cuda.synchronize()
for rnd in range(6):
    for t in ((16, 32) if rnd % 2 == 0 else (32, 16)):
        s = smp[t]
        for _ in range(5): s.plugin_sample(wi, None, T=T, seed=3, out=(wo, p))
        torch.cuda.synchronize(); s.set_profiling(True)
        for _ in range(40): s.plugin_sample(wi, None, T=T, seed=3, out=(wo, p))
        k, ms = s.profile_read(); mhz = s.profile_clock_mhz(); s.set_profiling(False)
        print(f"round {rnd} tile {t}: {ms / k * 1e3:7.1f} us per launch @ {mhz:6.0f} MHz = {ms / k * mhz:8.1f} kcycles", flush=True)
