#!/usr/bin/env python3
"""Per-launch fixed cost (prologue + epilogue + launch ramp) vs per-step cost of the plugin-level calls: kernel time (HIP events
on the launch stream) at T = 1, 2, 4, 8, 16 for sample() / pdf() with and without the per-query context and with an injected x0,
fitted as t = a + b T.   python tools/fixed_cost.py [disk|spherical] [N]"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, time
import bench
from bsdf_diffusion_sampling_amd import weights as W
from bsdf_diffusion_sampling_amd.sampler import FlowSampler
dom = sys.argv[1] if len(sys.argv) > 1 else "disk"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
dev = torch.device("cuda")
s = FlowSampler(W.load(W.shipped_path("aniso_miro_7_rgb", dom)))
wi = bench.make_wi(dom, n, 1234, dev)
wo = torch.empty((n, 3), device=dev); p = torch.empty(n, device=dev)
ctx = s.new_context(n)
x0 = torch.zeros((n, 2), device=dev) + 0.1
t0 = time.time()
while time.time() - t0 < 0.2:
    s.plugin_sample(wi, None, T=8, out=(wo, p)); torch.cuda.synchronize()
s.plugin_sample(wi, None, T=8, seed=3, out=(wo, p), ctx_out=ctx)
cases = {
    "sample (in-kernel draw)": lambda T: s.plugin_sample(wi, None, T=T, seed=3, out=(wo, p)),
    "sample + ctx write": lambda T: s.plugin_sample(wi, None, T=T, seed=3, out=(wo, p), ctx_out=ctx),
    "sample, ctx read": lambda T: s.plugin_sample(wi, None, T=T, seed=3, out=(wo, p), ctx_in=ctx),
    "sample, injected x0": lambda T: s.plugin_sample(wi, x0, T=T, seed=3, out=(wo, p)),
    "pdf": lambda T: s.plugin_pdf(wi, wo, T=T, out=p),
    "pdf, ctx read": lambda T: s.plugin_pdf(wi, wo, T=T, out=p, ctx_in=ctx),
    "pdf + ctx write": lambda T: s.plugin_pdf(wi, wo, T=T, out=p, ctx_out=ctx),
}
Ts = (1, 2, 4, 8, 16)
for name, fn in cases.items():
    res = []
    for T in Ts:
        for _ in range(3): fn(T)
        torch.cuda.synchronize(); s.set_profiling(True)
        for _ in range(10): fn(T)
        k, ms = s.profile_read(); res.append(ms / k)
        mhz = s.profile_clock_mhz(); s.set_profiling(False)
    b, a = np.polyfit(Ts, res, 1)
    print(f"{dom} {name:26s} " + " ".join(f"T{T}={v*1e3:6.1f}" for T, v in zip(Ts, res)) + f" us | per step {b*1e3:5.1f} us, fixed {a*1e3:5.1f} us ({a/(a+8*b)*100:4.1f} % at T=8, {a/(a+4*b)*100:4.1f} % at T=4) @{mhz:.0f} MHz", flush=True)
