#!/usr/bin/env python3
"""Time the REFERENCE's CPU PyTorch path against oracle/torch_eager_port.py on the same
inputs and cores (build container only; needs /root/reference).  Validates that the port
used for bench.py's ``cpu_baseline`` leg costs what the reference costs (BASELINE.md §4:
equal outputs, wall time within ~10 %).  Writes tests/golden/cpu_timing.json."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (sets up the reference import recipe)
import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, MG.ROOT)
from bsdf_diffusion_sampling_amd import weights as W  # noqa: E402
from oracle import torch_eager_port as P  # noqa: E402


def med_time(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def main():
    torch.set_num_threads(os.cpu_count())
    res = {"threads": os.cpu_count(), "torch": torch.__version__, "cases": []}
    N = 262144
    for mat, dom, T in (("aniso_miro_7_rgb", "disk", 4), ("aniso_miro_7_rgb", "disk", 8),
                        ("aniso_miro_7_rgb", "spherical", 8)):
        g = torch.Generator().manual_seed(1234)
        u = torch.rand(N, 2, generator=g)
        if dom == "disk":
            r, a = 0.95 * torch.sqrt(u[:, 0]), 2 * np.pi * u[:, 1]
            wi = torch.stack([r * torch.cos(a), r * torch.sin(a)], 1).float()
        else:
            wi = torch.stack([1.5 * u[:, 0], (2 * u[:, 1] - 1) * np.pi], 1).float()
        fw = W.load(W.shipped_path(mat, dom))
        base, net = P.BaseNet(fw), P.VelocityNet(fw)
        with MG.CudaToCpu():
            db, ds = MG.build_nets(mat, dom, None, torch.float32)
            fs = MG.ref_ops.network_sampling_disk if dom == "disk" else MG.ref_ops.network_sampling_spherical
            fp = MG.ref_ops.network_pdf_disk if dom == "disk" else MG.ref_ops.network_pdf_spherical
            with torch.no_grad():
                x0 = db.sample(wi, N).detach()
            db.sample = lambda c, n=1: x0.clone()
            xr, pr = fs(db, ds, wi, T=T)
            xp, pp = P.network_sampling(base, net, wi, T, x0=x0)
            t_ref_s = med_time(lambda: fs(db, ds, wi, T=T))
            t_ref_p = med_time(lambda: fp(db, ds, xr, wi, T=T))
        t_port_s = med_time(lambda: P.network_sampling(base, net, wi, T, x0=x0))
        t_port_p = med_time(lambda: P.network_pdf(base, net, xr, wi, T))
        res["cases"].append({
            "material": mat, "domain": dom, "T": T, "N": N,
            "x_max_abs_diff": float((xr - xp).abs().max()), "pdf_max_rel_diff": float(((pr - pp).abs() / pr.abs().clamp_min(1e-30)).max()),
            "ref_sample_s": t_ref_s, "port_sample_s": t_port_s, "ref_pdf_s": t_ref_p, "port_pdf_s": t_port_p,
            "ref_sample_Msps": N / t_ref_s / 1e6, "port_sample_Msps": N / t_port_s / 1e6,
            "ref_pdf_Msps": N / t_ref_p / 1e6, "port_pdf_Msps": N / t_port_p / 1e6})
        print(res["cases"][-1], flush=True)
    with open(os.path.join(HERE, "cpu_timing.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
