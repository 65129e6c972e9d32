#!/usr/bin/env python3
"""Secondary workloads (not the headline bench.py line):

  mixed    BASELINE.json configs[3] per-GPU share: all 52 measured materials (27 disk T=4 + 25
           spherical T=8), 16 Mi queries with a uniformly random material id, bucketed by id,
           ONE segmented launch per kernel signature (27 disk + 25 spherical materials = 2 launches) for sample() and 2 for pdf().
  teacher  the reference's only tiny-cuda-nn call site (reflow `dosampling`,
           learning_repo_cleanup/spherical_domain_sampling.py:147-166): 64-wide x 6 teacher net,
           4 Mi rows, T = 128 Euler steps, no Jacobian, fp16.
"""
import argparse, json, sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from bsdf_diffusion_sampling_amd import weights as W
from bsdf_diffusion_sampling_amd.materials import MaterialTable
from bsdf_diffusion_sampling_amd.sampler import FlowSampler


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=["mixed", "teacher"])
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    if a.workload == "mixed":
        n = 1 << 24
        tab = MaterialTable.all_measured()
        g = torch.Generator().manual_seed(1)
        ids = torch.randint(0, len(tab), (n,), generator=g).to(dev)
        wi = bench.make_wi("spherical", n, 1234, dev)  # unit vectors on the upper hemisphere serve both domains
        state = {}

        def step():
            plan = tab.bucket(ids)  # one stable sort per wavefront, shared by sample() and pdf()
            state["wo"], state["pdf"] = tab.sample(plan, wi, seed=5)
            state["p"] = tab.pdf(plan, wi, state["wo"])
        dt = timed(step, a.steps, a.warmup)
        flops = 0
        counts = torch.bincount(ids, minlength=len(tab)).cpu().tolist()
        for m, c in enumerate(counts):
            flops += 2 * c * tab.samplers[m].flops_per_query(tab.T[m])
        print(json.dumps({"workload": "mixed_52materials_16Mi", "Msamples_per_s": n / dt / 1e6, "ms_per_step": dt * 1e3,
                          "materials": len(tab), "algorithmic_TFLOPs": flops / dt / 1e12,
                          "note": "includes the bucketing sort (once per step) and the gather/scatter (torch) around 4 segmented launches (2 sample + 2 pdf)"}))
    else:
        n, T = 1 << 22, 128
        fw = W.load(W.shipped_path("aniso_miro_7_rgb", "spherical", "complex"))
        s = FlowSampler(fw, precision="f16")
        g = torch.Generator().manual_seed(2)
        u = torch.rand(n, 2, generator=g)
        cond = torch.stack([1.5 * u[:, 0], (2 * u[:, 1] - 1) * np.pi], 1).float().to(dev)
        x0 = torch.stack([0.7 + 0.3 * torch.randn(n, generator=g), (2 * torch.rand(n, generator=g) - 1) * np.pi], 1).float().to(dev)
        dt = timed(lambda: s.flow_samples_only(cond, x0, T=T), a.steps, a.warmup)
        w = fw.width
        fwd = 2 * (fw.in_dim * w + (fw.n_hidden - 1) * w * w + 2 * w) * T
        print(json.dumps({"workload": "teacher_64x6_4Mi_T128_samples_only_f16", "Mrows_per_s": n / dt / 1e6,
                          "ms_per_call": dt * 1e3, "algorithmic_TFLOPs": n * fwd / dt / 1e12,
                          "frac_fp16_mfma_peak": n * fwd / dt / 2.5e15}))


if __name__ == "__main__":
    main()
