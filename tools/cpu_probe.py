"""How long does ONE sample()+pdf() pass of the CPU eager port take on this host with ALL cores, by batch size?
(sizing of bench.py's cpu_baseline.all_cores sample).  python3 tools/cpu_probe.py [budget_s]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = """
import sys, time, numpy as np
sys.path.insert(0, {root!r})
import torch
from bsdf_diffusion_sampling_amd import weights as W
from oracle import torch_eager_port as P
torch.set_num_threads({thr})
fw = W.load(W.shipped_path("aniso_miro_7_rgb", "disk"))
base, net = P.BaseNet(fw), P.VelocityNet(fw)
g = torch.Generator().manual_seed(1234)
u = torch.rand({n}, 2, generator=g)
r, a = 0.95 * torch.sqrt(u[:, 0]), 2 * np.pi * u[:, 1]
cond = torch.stack([r * torch.cos(a), r * torch.sin(a)], 1).float()
for i in range(3):
    t0 = time.perf_counter()
    x, _ = P.network_sampling(base, net, cond, 8)
    P.network_pdf(base, net, x, cond, 8)
    print("PASS", i, time.perf_counter() - t0, flush=True)
"""
ncpu = len(os.sched_getaffinity(0))
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
print("host cpus:", ncpu, flush=True)
for thr in sorted({ncpu, max(ncpu // 2, 1), max(ncpu // 4, 1)}, reverse=True):
    for n in (256, 2048, 16384):
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, "-c", CODE.format(root=ROOT, thr=thr, n=n)], capture_output=True, text=True, timeout=budget)
            txt = r.stdout
        except subprocess.TimeoutExpired as e:
            txt = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        print(f"threads={thr} n={n}: wall {time.perf_counter() - t0:.1f}s ->", [float(l.split()[2]) for l in txt.splitlines() if l.startswith("PASS")], flush=True)
