import sys, os
sys.path.insert(0, '.')
import numpy as np, torch, time
import bench
from bsdf_diffusion_sampling_amd import weights as W
from bsdf_diffusion_sampling_amd.sampler import FlowSampler
dom = "spherical"; n = 1 << 20
dev = torch.device("cuda")
s = FlowSampler(W.load(W.shipped_path("aniso_miro_7_rgb", dom)))
wi = bench.make_wi(dom, n, 1234, dev)
wo = torch.empty((n, 3), device=dev); p = torch.empty(n, device=dev)
t0 = time.time()
while time.time() - t0 < 0.2:
    s.plugin_pdf(wi, wi, T=8, out=p); torch.cuda.synchronize()
wo2 = bench.make_wi(dom, n, 77, dev)
out = {}
for name, fn in {"sample": lambda T: s.plugin_sample(wi, None, T=T, seed=3, out=(wo, p)), "pdf": lambda T: s.plugin_pdf(wi, wo2, T=T, out=p)}.items():
    res = []
    for T in (1, 8):
        for _ in range(3): fn(T)
        torch.cuda.synchronize(); s.set_profiling(True)
        for _ in range(15): fn(T)
        k, ms = s.profile_read(); res.append(ms / k); s.set_profiling(False)
    out[name] = res
print(os.environ.get("BSDFD_LIB_PATH", "").split("lib_")[-1], " ".join(f"{k}: T1 {v[0]*1e3:6.1f} T8 {v[1]*1e3:6.1f} us |" for k, v in out.items()), flush=True)
