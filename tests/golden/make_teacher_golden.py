#!/usr/bin/env python3
"""Golden vectors for the DISK reflow teacher sampler, produced by running the reference's model.

The reference's `dosampling` (learning_repo_cleanup/disk_domain_sampling.py:93-110) is a closure inside its training
script and evaluates the teacher through tiny-cuda-nn (CUDA only), so it cannot be called here; what it computes is
    x <- x + 1/T * D(x, t/T, omega_i),  t = 0..T-1,   D = `brdf_diffusion_network<material>.pth` (32 x 3, no Jacobian)
and `D` as a PyTorch module is the reference's own `NN_cond_pos_simpler` (disk_domain_sampling.py:73-75 loads the same
pickle into it).  This script runs THAT module, imported in place from /root/reference (recipe of make_golden.py), in
fp32 and fp64, for T = 128 and the script's default T = 256 (:151), from stored base draws x0.

Usage:  python tests/golden/make_teacher_golden.py      (build container only; writes disk_teacher_*.npz next to itself)
"""
import os

import numpy as np
import torch

import make_golden as G  # sets up the import of the reference's rendering/utils/model.py

MATERIAL = "aniso_miro_7_rgb"
N = 1024


def run():
    d = os.path.join(G.CKPT, f"{MATERIAL}_disk")
    sd = torch.load(os.path.join(d, f"brdf_diffusion_network{MATERIAL}.pth"), map_location="cpu")
    out = {}
    g = torch.Generator().manual_seed(41)
    u = torch.rand(N, 2, generator=g)
    r, a = 0.95 * torch.sqrt(u[:, 0]), 2 * np.pi * u[:, 1]
    cond = torch.stack([r * torch.cos(a), r * torch.sin(a)], 1).float()
    x0 = (0.35 * torch.randn(N, 2, generator=g)).float()
    out["wi"], out["x0"] = cond.numpy(), x0.numpy()
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        torch.set_default_dtype(dtype)
        net = G.ref_model.NN_cond_pos_simpler(input_dim=5, output_dim=2, N_NEURONS=32, POSITIONAL_ENCODING_BASIS_NUM=5)
        net.load_state_dict(sd)
        net = net.to(dtype).eval()
        for T in (128, 256):
            x = x0.to(dtype).clone()
            c = cond.to(dtype)
            ones = torch.ones(N, 1, dtype=dtype)
            with torch.no_grad():
                for t in range(T):
                    x = x + 1 / T * net(x, t / T * ones, c)     # disk_domain_sampling.py:104-108
            out[f"x_T{T}_{tag}"] = x.numpy()
    torch.set_default_dtype(torch.float32)
    return out


if __name__ == "__main__":
    res = run()
    np.savez_compressed(os.path.join(G.HERE, f"disk_teacher_{MATERIAL}.npz"), **res)
    print({k: v.shape for k, v in res.items()})
    print("fp32 vs fp64 at T=256:", np.abs(res["x_T256_f32"] - res["x_T256_f64"]).max())
    assert not os.path.exists(os.path.join(G.REF, "utils", "__pycache__")), "wrote into the reference tree"
