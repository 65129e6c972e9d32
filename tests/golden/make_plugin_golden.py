#!/usr/bin/env python3
"""Plugin-level golden vectors (last row: "(wo3, pdf_sa) after guards").

The reference's plugin classes need Mitsuba + DrJit (absent from the image), so they cannot be imported.  What they
do around the operators is a handful of tensor ops; this script applies exactly those ops, in fp32 torch like the
plugins, to the outputs of the reference's OWN operators (``network_sampling_*`` / ``network_pdf_*`` imported in place
from /root/reference by ``make_golden.py``'s recipe) and stores inputs and results next to the operator-level
fixtures as ``<stem>_plugin.npz``.  Each block below names the plugin lines it applies:

  disk      rendering/brdf_measured_disk.py:59-82 (sample), :112-124 (pdf), utils/mitsuba_brdf_draw.py:40-43
  spherical rendering/brdf_measured_spherical.py:35-39 (cart_to_spher), :30-33 (sph_to_dir), :69-91, :122-137
  full      rendering/bsdf_myresult.py:59-84, :115-133 (no cos masks, |sin theta|, no sin guard in pdf)

DrJit-side ops used by those lines are Frame3f.cos_theta (= z), Frame3f.sin_theta (= sqrt(x^2 + y^2)), dr.sincos,
dr.clamp and dr.select — evaluated here with the fp32 torch equivalents.  The measured.eval()-dependent firefly rule
(:97-100 / :106-108 / :100-103) is NOT part of these fixtures (Mitsuba's `measured` plugin cannot run here).

Also stored: the same post-processing applied to the reference's fp64 run of the sampling operator (``*_f64`` keys; the
pdf operators force fp32 at mlp_brdf_sampling.py:71,146, so there is no fp64 pdf run) — it shows how much of a plugin-level
discrepancy is the reference's own fp32 arithmetic (acos / atan2 of cart_to_spher, the flow itself).

Usage:  python tests/golden/make_plugin_golden.py      (build container only; nothing is written under /root/reference)
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (stubs, TorchFunctionMode, build_nets; imports the reference in place)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ref_ops = G.ref_ops
FLT_MAX = float(np.finfo(np.float32).max)


def cart_to_spher(xyz):  # rendering/brdf_measured_spherical.py:35-39, verbatim semantics
    r = torch.norm(xyz, dim=1)
    theta = torch.acos(xyz[:, 2] / (r + 1e-8))
    phi = torch.atan2(xyz[:, 1], xyz[:, 0])
    return torch.stack([theta, phi], dim=1)


def sph_to_dir(theta, phi):  # :30-33 (dr.sincos)
    st, ct = torch.sin(theta), torch.cos(theta)
    sp, cp = torch.sin(phi), torch.cos(phi)
    return torch.stack([cp * st, sp * st, ct], dim=1)


def sin_theta(v):  # mi.Frame3f.sin_theta: safe_sqrt(x^2 + y^2)
    return torch.sqrt(torch.clamp(v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1], min=0))


def disk_sample(db, ds, wi3, T):
    wo, pdf = ref_ops.network_sampling_disk(db, ds, wi3[:, :2], T=T)          # :66-68
    wo, pdf = wo.detach().clone(), pdf.detach().clone()
    valid = torch.square(wo[..., 0]) + torch.square(wo[..., 1]) < 0.995        # :69
    wo[~valid] = torch.tensor([0.0, 0.0], dtype=wo.dtype)                      # :70
    pdf[~valid] = 0.0                                                          # :71
    rr = wo[..., :2].pow(2).sum(-1)                                            # disk_to_cart, mitsuba_brdf_draw.py:40-43
    wo3 = torch.cat([wo, (1 - rr).relu().sqrt().unsqueeze(-1)], -1)
    return wo3, pdf * wo3[:, 2]                                                # :81-82  bs.pdf = pdf * cos_theta_o


def disk_pdf(db, ds, wi3, wo3, T):
    pdf = ref_ops.network_pdf_disk(db, ds, wo3[:, :2], wi3[:, :2], T=T).detach()   # :116-120
    ok = (wi3[:, 2] > 0.0) & (wo3[:, 2] > 0.0)                                      # :122-124
    return torch.where(ok, pdf * wo3[:, 2], torch.zeros_like(pdf))


def spherical_sample(db, ds, wi3, T, full):
    wi_in = cart_to_spher(wi3)                                                  # :76-77
    wo, pdf = ref_ops.network_sampling_spherical(db, ds, wi_in, T=T)            # :78
    wo, pdf = wo.detach(), pdf.detach()
    pdf = torch.where(torch.sin(wo[:, 0]) > 0.00005, pdf, torch.zeros_like(pdf))        # :79
    if not full:
        pdf = torch.where(torch.cos(wo[:, 0]) > 0, pdf, torch.zeros_like(pdf))          # :80 (absent in bsdf_myresult.py)
    wo3 = sph_to_dir(wo[:, 0], wo[:, 1])                                                # :81-82
    s = sin_theta(wo3)
    inv = torch.clamp(1 / (s.abs() if full else s), 1, FLT_MAX)                         # :89 / bsdf_myresult.py:80
    return wo3, pdf * inv, wo


def spherical_pdf(db, ds, wi3, wo3, T, full):
    wi_in, wo_in = cart_to_spher(wi3), cart_to_spher(wo3)                               # :128-131
    pdf = ref_ops.network_pdf_spherical(db, ds, wo_in, wi_in, T=T).detach()             # :132
    s = sin_theta(wo3)
    inv = torch.clamp(1 / (s.abs() if full else s), 1, FLT_MAX)
    if full:                                                                            # bsdf_myresult.py:115-133
        return pdf * inv
    pdf = torch.where(torch.sin(wo_in[:, 0]) > 0.00005, pdf, torch.zeros_like(pdf))     # :133
    ok = (wi3[:, 2] > 0.0) & (wo3[:, 2] > 0.0)                                          # :134-136
    return torch.where(ok, pdf * inv, torch.zeros_like(pdf))


def run_case(material, domain, variant, seed, suffix=""):
    stem = f"{material}_{domain}" + (f"_{variant}" if variant else "") + suffix
    g = np.load(os.path.join(HERE, stem + ".npz"))
    G.N = g["wi"].shape[0]
    T = int(g["meta_T"])
    full = material.startswith("bsdf_")
    wi = torch.from_numpy(g["wi"])
    x0 = torch.from_numpy(g["x0"])
    out = {"meta_material": material, "meta_domain": domain, "meta_variant": variant or "", "meta_T": T,
           "meta_full_sphere": int(full), "meta_torch": torch.__version__}
    rng = torch.Generator().manual_seed(seed + 100)
    if domain == "disk":
        wi3 = torch.cat([wi, torch.sqrt(torch.clamp(1 - (wi * wi).sum(1, keepdim=True), min=0))], 1).float()
        x0 = x0.clone()
        x0[:16] *= 40.0          # some draws leave the disk: the r^2 >= 0.995 guard
    else:
        wi3 = sph_to_dir(wi[:, 0].double(), wi[:, 1].double()).float()   # unit vectors as a renderer hands them over
    # directions whose pdf is asked: fresh ones over the (upper | full) sphere, some on the wrong side of the masks
    z = torch.rand(G.N, generator=rng) * (2.0 if full else 1.0) - (1.0 if full else 0.0)
    ph = (2 * torch.rand(G.N, generator=rng) - 1) * np.pi
    r = torch.sqrt(torch.clamp(1 - z * z, min=0))
    wl3 = torch.stack([r * torch.cos(ph), r * torch.sin(ph), z], 1).float()
    wl3[:8, 2] *= -1.0           # cos(theta_o) <= 0 lanes
    wi3m = wi3.clone()
    wi3m[8:16, 2] *= -1.0        # cos(theta_i) <= 0 lanes
    out.update(wi3=wi3.numpy(), x0=x0.numpy(), pdf_wi3=wi3m.numpy(), pdf_wo3=wl3.numpy())
    with G.CudaToCpu():
        torch.set_default_dtype(torch.float32)
        db, ds = G.build_nets(material, domain, variant, torch.float32)
        db.sample = lambda cond, n=1, _x0=x0: _x0.clone()
        if domain == "disk":
            wo3, pdf_sa = disk_sample(db, ds, wi3, T)
            out.update(sample_wo3=wo3.numpy(), sample_pdf_sa=pdf_sa.numpy())
            out["pdf_sa"] = disk_pdf(db, ds, wi3m, wl3, T).numpy()
            out["pdf_sa_of_samples"] = disk_pdf(db, ds, wi3, wo3, T).numpy()
        else:
            wo3, pdf_sa, wo2 = spherical_sample(db, ds, wi3, T, full)
            out.update(sample_wo3=wo3.numpy(), sample_pdf_sa=pdf_sa.numpy(), sample_theta_phi=wo2.numpy(),
                       wi_theta_phi=cart_to_spher(wi3).numpy())
            out["pdf_sa"] = spherical_pdf(db, ds, wi3m, wl3, T, full).numpy()
            out["pdf_sa_of_samples"] = spherical_pdf(db, ds, wi3, wo3, T, full).numpy()
        # the reference's own fp64 run of sample() with the same post-processing in fp64
        torch.set_default_dtype(torch.float64)
        db64, ds64 = G.build_nets(material, domain, variant, torch.float64)
        x0d = x0.double()
        db64.sample = lambda cond, n=1, _x0=x0d: _x0.clone()
        if domain == "disk":
            wo3d, pdfd = disk_sample(db64, ds64, wi3.double(), T)
        else:
            wo3d, pdfd, _ = spherical_sample(db64, ds64, wi3.double(), T, full)
        out.update(sample_wo3_f64=wo3d.numpy(), sample_pdf_sa_f64=pdfd.numpy())
        torch.set_default_dtype(torch.float32)
    return stem, out


if __name__ == "__main__":
    if "--large" in sys.argv[1:]:   # the hard case at 16 384 rows (tests/golden/make_golden.py --large first)
        for mat, dom, var, seed in G.LARGE_CASES:
            if var:
                continue
            stem, res = run_case(mat, dom, var, seed, suffix="_n16k")
            res = {k: v for k, v in res.items() if k not in ("sample_theta_phi", "wi_theta_phi")}
            np.savez_compressed(os.path.join(HERE, stem + "_plugin.npz"), **res)
            print("wrote", stem + "_plugin", {k: v.shape for k, v in res.items() if hasattr(v, "shape") and v.ndim}, flush=True)
        assert not os.path.exists(os.path.join(G.REF, "utils", "__pycache__")), "wrote into the reference tree"
        sys.exit(0)
    for mat, dom, var, seed in G.CASES:
        if var:  # the 64-wide teacher is not loaded by any plugin
            continue
        stem, res = run_case(mat, dom, var, seed)
        np.savez_compressed(os.path.join(HERE, stem + "_plugin.npz"), **res)
        print("wrote", stem + "_plugin", {k: v.shape for k, v in res.items() if hasattr(v, "shape") and v.ndim}, flush=True)
    assert not os.path.exists(os.path.join(G.REF, "utils", "__pycache__")), "wrote into the reference tree"
    print("done")
