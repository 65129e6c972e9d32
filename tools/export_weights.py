#!/usr/bin/env python3
"""Convert the reference's ``checkpoints_new/`` pickles into neutral ``.bsdfw`` files.

Run once where torch and the reference checkpoints are available (this
container); the output under bsdf_diffusion_sampling_amd/data/weights/ is pure
fp32 data and is what the GPU box uses.  Only the nets the plugins load are
converted (rendering/brdf_measured_disk.py:43-51, brdf_measured_spherical.py:53-59,
bsdf_myresult.py:49-54): ``brdf_rectify_network*`` + ``brdf_pretrain_network*``;
``--complex a,b`` additionally converts the 64-wide 6-hidden teachers of the named materials
(``brdf_diffusion_network_complex*``, rendering/utils/model.py:449-477).

NOTE the spherical plugin of the reference loads the ``_disk`` pretrain net into
the spherical base (brdf_measured_spherical.py:59) — a reference bug (SURVEY.md
§0); this exporter pairs ``_spherical`` with ``_spherical``.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bsdf_diffusion_sampling_amd import weights as W  # noqa: E402


def _load(p):
    return torch.load(p, map_location="cpu")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", default="/root/reference/rendering/checkpoints_new")
    ap.add_argument("--out", default=W.DATA_DIR)
    ap.add_argument("--complex", default="aniso_miro_7_rgb,chm_orange_rgb,bsdf_3",
                    help="comma list of materials whose 64-wide teacher net is exported too")
    ap.add_argument("--diffusion", default="aniso_miro_7_rgb",
                    help="comma list of materials whose DISK reflow teacher (brdf_diffusion_network*, 32 x 3: "
                         "learning_repo_cleanup/disk_domain_sampling.py:73-75) is exported too, as <mat>_disk_diffusion.bsdfw")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    want_diffusion = set(filter(None, a.diffusion.split(",")))
    n_diff = 0
    n = 0
    n_complex = 0
    want_complex = set(filter(None, a.complex.split(",")))
    for d in sorted(os.listdir(a.ckpt)):
        full = os.path.join(a.ckpt, d)
        if not os.path.isdir(full):
            continue
        if d.endswith("_disk"):
            mat, dom, domain = d[:-5], "disk", W.DOMAIN_DISK
        elif d.endswith("_spherical"):
            mat, dom, domain = d[:-10], "spherical", W.DOMAIN_SPHERICAL
        else:
            continue
        tag = mat[5:] if mat.startswith("bsdf_") else mat  # bsdf_<i> files are named by index
        rect = os.path.join(full, f"brdf_rectify_network{tag}.pth")
        pre = os.path.join(full, f"brdf_pretrain_network{tag}.pth")
        if not (os.path.exists(rect) and os.path.exists(pre)):
            print(f"skip {d}: missing rectify/pretrain")
            continue
        base_sd = _load(pre)
        fw = W.from_state_dicts(mat, domain, _load(rect), base_sd)
        W.save(os.path.join(a.out, f"{mat}_{dom}.bsdfw"), fw)
        n += 1
        dif = os.path.join(full, f"brdf_diffusion_network{tag}.pth")
        if dom == "disk" and os.path.exists(dif) and mat in want_diffusion:
            W.save(os.path.join(a.out, f"{mat}_{dom}_diffusion.bsdfw"), W.from_state_dicts(mat, domain, _load(dif), base_sd))
            n_diff += 1
        cpx = os.path.join(full, f"brdf_diffusion_network_complex{tag}.pth")
        if dom == "spherical" and os.path.exists(cpx) and mat in want_complex:
            fwc = W.from_state_dicts(mat, domain, _load(cpx), base_sd)
            W.save(os.path.join(a.out, f"{mat}_{dom}_complex.bsdfw"), fwc)
            n_complex += 1
    print(f"wrote {n} weight sets (+{n_complex} complex, +{n_diff} disk teachers) to {a.out}")


if __name__ == "__main__":
    main()
