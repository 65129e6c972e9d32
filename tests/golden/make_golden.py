#!/usr/bin/env python3
"""Generate the committed golden vectors by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference and torch on CPU); the
``.npz`` files it writes next to itself are data (inputs + the reference's
outputs) and are what travels to the GPU box.  Nothing of the reference's source
is copied: its modules are imported in place, unmodified, with the recipe of
SURVEY.md §8(c):

  * ``sys.dont_write_bytecode`` so nothing is written under /root/reference;
  * empty stub modules for imageio / OpenEXR / Imath (pulled in by
    rendering/utils/utils.py via rendering/utils/mlp_brdf_sampling.py:9);
  * a ``TorchFunctionMode`` that rewrites the hard-coded ``device='cuda'`` /
    ``.to("cuda", ...)`` of mlp_brdf_sampling.py:21-23,27,71 to CPU;
  * ``D_base.sample`` monkey-patched to return a stored x0, so the deterministic
    part x0 -> (x_T, pdf) is bit-reproducible (the torch RNG stream is not).

Usage:  python tests/golden/make_golden.py            # the 2 048-row fixtures <stem>.npz (+ kappa sweep, toy)
        python tests/golden/make_golden.py --large    # 16 384-row fixtures <stem>_n16k.npz of the two HARD cases (LARGE_CASES):
                                                      # the per-row arrays of the operators only, so that a p99 has 160 rows
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference/rendering"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

for _m in ("imageio", "OpenEXR", "Imath"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
import matplotlib  # noqa: E402

matplotlib.use("Agg")
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.overrides import TorchFunctionMode  # noqa: E402

sys.path.insert(0, REF)
sys.path.insert(1, ROOT)
import utils.model as ref_model  # noqa: E402  (the reference's rendering/utils/model.py)
import utils.mlp_brdf_sampling as ref_ops  # noqa: E402

CKPT = os.path.join(REF, "checkpoints_new")
N = 2048
N_STEP = 64


class CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if "device" in kwargs and str(kwargs["device"]).startswith("cuda"):
            kwargs["device"] = "cpu"
        args = tuple("cpu" if (isinstance(a, str) and a.startswith("cuda")) else a for a in args)
        return func(*args, **kwargs)


def build_nets(material, domain, variant, dtype):
    tag = material[5:] if material.startswith("bsdf_") else material
    d = os.path.join(CKPT, f"{material}_{domain}")
    if domain == "disk":
        # rendering/brdf_measured_disk.py:43,49
        ds = ref_model.NN_cond_pos_simpler(input_dim=5, output_dim=2, N_NEURONS=32,
                                           POSITIONAL_ENCODING_BASIS_NUM=5)
        db = ref_model.NN_cond_pretrain_disk_one(input_dim=2, N_NEURONS=16,
                                                 POSITIONAL_ENCODING_BASIS_NUM=3)
        sname = f"brdf_rectify_network{tag}.pth"
    else:
        if variant == "complex":
            ds = ref_model.NN_cond_pos_spherical_complicate(input_dim=6, output_dim=2, N_NEURONS=64,
                                                           POSITIONAL_ENCODING_BASIS_NUM=5)
            sname = f"brdf_diffusion_network_complex{tag}.pth"
        else:
            # rendering/brdf_measured_spherical.py:53,58
            ds = ref_model.NN_cond_pos(input_dim=6, output_dim=2, N_NEURONS=32,
                                       POSITIONAL_ENCODING_BASIS_NUM=5)
            sname = f"brdf_rectify_network{tag}.pth"
        db = ref_model.NN_cond_pretrain_spherical_one(input_dim=2, N_NEURONS=16)
    ds.load_state_dict(torch.load(os.path.join(d, sname), map_location="cpu"))
    db.load_state_dict(torch.load(os.path.join(d, f"brdf_pretrain_network{tag}.pth"),
                                  map_location="cpu"))
    return db.to(dtype).eval(), ds.to(dtype).eval()


def make_inputs(domain, seed):
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(N, 2, generator=g)
    if domain == "disk":
        # uniform on the disk of radius 0.95
        r, a = 0.95 * torch.sqrt(u[:, 0]), 2 * np.pi * u[:, 1]
        return torch.stack([r * torch.cos(a), r * torch.sin(a)], 1).float()
    # config 3: theta_i ~ U(0,1.5), phi_i ~ U(-pi,pi)
    return torch.stack([1.5 * u[:, 0], (2 * u[:, 1] - 1) * np.pi], 1).float()


def single_step(ds, domain, x, alpha, cond):
    """v and dv/dx through the reference net by autograd (as mlp_brdf_sampling.py:29-41)."""
    x = x.clone().requires_grad_(True)
    a = torch.full((x.shape[0], 1), alpha, dtype=x.dtype)
    if domain == "disk":
        v = ds(x, a, cond)
    else:
        per = torch.cat([torch.sin(x[:, 1]).reshape(-1, 1), torch.cos(x[:, 1]).reshape(-1, 1)], 1)
        v = ds(torch.cat([x[:, 0].reshape(-1, 1), per], 1), a, cond)
    g0 = torch.autograd.grad(v[:, 0].sum(), x, retain_graph=True)[0]  # grad of v_0
    g1 = torch.autograd.grad(v[:, 1].sum(), x)[0]                     # grad of v_1
    return v.detach(), g0, g1


def run_case(material, domain, variant, seed):
    T = 4 if domain == "disk" else 8
    out = {"meta_material": material, "meta_domain": domain, "meta_variant": variant or "",
           "meta_seed": seed, "meta_T": T, "meta_torch": torch.__version__}
    fn_s = ref_ops.network_sampling_disk if domain == "disk" else ref_ops.network_sampling_spherical
    fn_p = ref_ops.network_pdf_disk if domain == "disk" else ref_ops.network_pdf_spherical
    wi = make_inputs(domain, seed)
    out["wi"] = wi.numpy()
    with CudaToCpu():
        torch.set_default_dtype(torch.float32)
        db, ds = build_nets(material, domain, variant, torch.float32)
        torch.manual_seed(seed)
        with torch.no_grad():
            x0 = db.sample(wi, wi.shape[0]).detach()
        out["x0"] = x0.numpy()
        out["pe5_rows"] = ref_model.positional_encoding_1(wi[:N_STEP], 5).numpy()
        out["pe3_rows"] = ref_model.positional_encoding_1(wi[:N_STEP], 3).numpy()
        db.sample = lambda cond, n=1, _x0=x0: _x0.clone()
        # --- a7 / a9 in fp32, several T
        for TT in sorted({T, 8, 1}):
            xs, ps = fn_s(db, ds, wi, T=TT)
            out[f"sample_x_T{TT}"] = xs.detach().numpy()
            out[f"sample_pdf_T{TT}"] = ps.detach().numpy()
        with torch.no_grad():
            out["base_logp_x0"] = db.log_prob(x0, wi).numpy()
            out["base_fwd"] = db.forward(wi).numpy()
        # --- a8 / a10: omega_o = the produced samples, plus fresh random outgoing points
        wo_a = torch.from_numpy(out[f"sample_x_T{T}"]).clone()
        g = torch.Generator().manual_seed(seed + 1)
        if domain == "disk":
            u = torch.rand(N, 2, generator=g)
            r, a = 0.99 * torch.sqrt(u[:, 0]), 2 * np.pi * u[:, 1]
            wo_b = torch.stack([r * torch.cos(a), r * torch.sin(a)], 1).float()
        else:
            u = torch.rand(N, 2, generator=g)
            hi = 3.1 if material.startswith("bsdf_") else 1.55
            wo_b = torch.stack([hi * u[:, 0], (2 * u[:, 1] - 1) * np.pi], 1).float()
        out["pdf_wo_a"], out["pdf_wo_b"] = wo_a.numpy(), wo_b.numpy()
        for TT in sorted({T, 8}):
            out[f"pdf_a_T{TT}"] = fn_p(db, ds, wo_a, wi, T=TT).detach().numpy()
            out[f"pdf_b_T{TT}"] = fn_p(db, ds, wo_b, wi, T=TT).detach().numpy()
        # --- single step v, grad v on a few rows, at two alphas
        for k, al in enumerate((0.0, 0.625)):
            v, g0, g1 = single_step(ds, domain, x0[:N_STEP], al, wi[:N_STEP])
            out[f"step{k}_alpha"] = np.float32(al)
            out[f"step{k}_v"], out[f"step{k}_g0"], out[f"step{k}_g1"] = v.numpy(), g0.numpy(), g1.numpy()
        # --- fp64 run of the same reference code (sampling only; the pdf functions
        #     force fp32 at mlp_brdf_sampling.py:71,146)
        torch.set_default_dtype(torch.float64)
        db64, ds64 = build_nets(material, domain, variant, torch.float64)
        x0d = x0.double()
        db64.sample = lambda cond, n=1, _x0=x0d: _x0.clone()
        xs, ps = fn_s(db64, ds64, wi.double(), T=T)
        out["sample_x_f64"], out["sample_pdf_f64"] = xs.detach().numpy(), ps.detach().numpy()
        torch.set_default_dtype(torch.float32)
    return out


def kappa_sweep():
    """torch VonMises.log_prob over concentrations crossing 3.75 (model.py:314)."""
    kappa = torch.tensor([1e-3, 0.01, 0.5, 1.0, 2.0, 3.0, 3.7, 3.749, 3.75, 3.751, 4.0, 8.0, 20.0,
                          50.0, 200.0, 1e3], dtype=torch.float32)
    phi = torch.linspace(-3.0, 3.0, 7)
    mu = torch.tensor(0.3)
    lp = torch.stack([torch.distributions.von_mises.VonMises(mu, k).log_prob(phi) for k in kappa])
    return {"kappa": kappa.numpy(), "phi": phi.numpy(), "mu": np.float32(0.3), "logp": lp.numpy()}


def toy_1d():
    """Config 1 plumbing: the reference's 1-D ``NN`` (model.py:78-98), random init
    under manual_seed(0); T=8 Euler steps with scalar Jacobian by autograd."""
    torch.manual_seed(0)
    net = ref_model.NN(input_dim=2, output_dim=1)
    x = torch.randn(16384, 1)
    out = {"x0": x.numpy().copy()}
    for i, lin in enumerate([net.linear1, net.linear2, net.linear3, net.linear4, net.output]):
        out[f"W{i}"], out[f"b{i}"] = lin.weight.detach().numpy(), lin.bias.detach().numpy()
    acc = torch.ones(x.shape[0])
    T = 8
    for t in range(T):
        x = x.detach().requires_grad_(True)
        v = net(x, torch.full_like(x, t / T))
        dv = torch.autograd.grad(v.sum(), x)[0]
        acc = acc / (1 + dv[:, 0] / T)
        x = x + v / T
    out["xT"], out["acc"] = x.detach().numpy()[:, 0], acc.detach().numpy()
    return out


CASES = [
    ("aniso_miro_7_rgb", "disk", None, 11),
    ("chm_orange_rgb", "disk", None, 12),
    ("vch_silk_blue_rgb", "disk", None, 13),
    ("aniso_miro_7_rgb", "spherical", None, 21),
    ("chm_orange_rgb", "spherical", None, 22),
    ("bsdf_3", "spherical", None, 23),
    ("aniso_miro_7_rgb", "spherical", "complex", 31),
]

LARGE_N = 16384
LARGE_CASES = [c for c in CASES if (c[0], c[1], c[2]) in (("chm_orange_rgb", "spherical", None), ("aniso_miro_7_rgb", "spherical", "complex"))]
LARGE_KEYS = ("wi", "x0", "pdf_wo_a", "pdf_wo_b", "sample_x_f64", "sample_pdf_f64")


def run_large():
    """The hard cases again at LARGE_N rows (same seeds, same code path: only N differs)."""
    global N
    N = LARGE_N
    for mat, dom, var, seed in LARGE_CASES:
        res = run_case(mat, dom, var, seed)
        T = res["meta_T"]
        keep = {k: v for k, v in res.items() if k.startswith("meta_") or k in LARGE_KEYS or
                k in (f"sample_x_T{T}", f"sample_pdf_T{T}", f"pdf_a_T{T}", f"pdf_b_T{T}")}
        stem = f"{mat}_{dom}" + (f"_{var}" if var else "") + "_n16k"
        np.savez_compressed(os.path.join(HERE, stem + ".npz"), **keep)
        print("wrote", stem, {k: v.shape for k, v in keep.items() if hasattr(v, "shape") and v.ndim}, flush=True)
    assert not os.path.exists(os.path.join(REF, "utils", "__pycache__")), "wrote into the reference tree"


if __name__ == "__main__":
    if "--large" in sys.argv[1:]:
        run_large()
        sys.exit(0)
    for mat, dom, var, seed in CASES:
        res = run_case(mat, dom, var, seed)
        stem = f"{mat}_{dom}" + (f"_{var}" if var else "")
        np.savez_compressed(os.path.join(HERE, stem + ".npz"), **res)
        print("wrote", stem, {k: v.shape for k, v in res.items() if hasattr(v, "shape") and v.ndim}, flush=True)
    np.savez_compressed(os.path.join(HERE, "von_mises_kappa_sweep.npz"), **kappa_sweep())
    np.savez_compressed(os.path.join(HERE, "toy_1d.npz"), **toy_1d())
    assert not os.path.exists(os.path.join(REF, "utils", "__pycache__")), "wrote into the reference tree"
    print("done")
