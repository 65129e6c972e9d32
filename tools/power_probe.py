#!/usr/bin/env python3
"""What the chip reports (rocm-smi: socket power, shader clock, temperature) while one kernel family runs back to back for a few
seconds:  python tools/power_probe.py [disk|spherical] [T]"""
import subprocess
import sys
import threading
import time
sys.path.insert(0, '.')
import torch
import bench
from bsdf_diffusion_sampling_amd import weights as W
from bsdf_diffusion_sampling_amd.sampler import FlowSampler

dom = sys.argv[1] if len(sys.argv) > 1 else "disk"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = 1 << 20
dev = torch.device("cuda")
fw = W.load(W.shipped_path("aniso_miro_7_rgb", dom))
wi = bench.make_wi(dom, n, 1234, dev)
wo = torch.empty((n, 3), device=dev); p = torch.empty(n, device=dev)


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=20)
        return r.stdout.strip()[:1500] or r.stderr.strip()[:300]
    except Exception as exc:
        return repr(exc)


print("idle:", smi(), flush=True)
for tile in (16, 32, 16, 32):
    s = FlowSampler(fw, tile=tile)
    stop = False
    out = []

    def sampler():
        time.sleep(1.0)
        while not stop:
            out.append(smi())
            time.sleep(0.5)
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    k = 0
    s.set_profiling(True)
    while time.time() - t0 < 4.0:
        for _ in range(20):
            s.plugin_sample(wi, None, T=T, seed=k, out=(wo, p)); k += 1
        torch.cuda.synchronize()
    nl, ms = s.profile_read(); mhz = s.profile_clock_mhz(); s.set_profiling(False)
    stop = True
    th.join()
    print(f"tile {tile}: {ms / nl * 1e3:.1f} us per launch @ {mhz:.0f} MHz in-kernel; rocm-smi samples:", flush=True)
    for o in out[:4]:
        print("   ", o.replace("\n", " ")[:900], flush=True)
    s.close()
