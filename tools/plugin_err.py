#!/usr/bin/env python3
"""Plugin-level error figures of ONE build / tile (BSDFD_LIB_PATH, BSDFD_TILE) on the six plugin fixtures of
tests/golden/*_plugin.npz — the numbers tests/test_gpu_parity.py::test_plugin_level_vs_reference_plugin_goldens asserts on.
Prints one JSON line:  {tag, <stem>: {sample_pdf_p99, sample_pdf_max, wo_max, pdf_p99, pdf_of_samples_p99}}"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bsdf_diffusion_sampling_amd import _lib  # noqa: E402
from bsdf_diffusion_sampling_amd.sampler import FlowSampler  # noqa: E402
from conftest import GOLDEN, load_case  # noqa: E402
from oracle import bsdf_oracle as O  # noqa: E402

CASES = ["aniso_miro_7_rgb_disk", "chm_orange_rgb_disk", "vch_silk_blue_rgb_disk", "aniso_miro_7_rgb_spherical",
         "chm_orange_rgb_spherical", "bsdf_3_spherical"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="lib")
    ap.add_argument("--tile", type=int, default=0)
    a = ap.parse_args()
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()  # noqa: E731
    rel = lambda u, v: np.abs(u - v) / np.maximum(np.abs(v), 1e-30)  # noqa: E731
    out = {"tag": a.tag}
    for stem in CASES:
        _, fw = load_case(stem)
        p = np.load(os.path.join(GOLDEN, stem + "_plugin.npz"))
        s = FlowSampler(fw, precision="split3", tile=a.tile)
        T, full = int(p["meta_T"]), bool(p["meta_full_sphere"])
        variant = _lib.PLUGIN_FULLSPHERE if full else _lib.PLUGIN_MEASURED
        wo, pdf = s.plugin_sample(t(p["wi3"]), t(p["x0"]), T=T, variant=variant)
        wo, pdf = wo.cpu().numpy().astype(np.float64), pdf.cpu().numpy().astype(np.float64)
        ref_wo, ref_pdf = p["sample_wo3_f64"], p["sample_pdf_sa_f64"]
        ok = np.abs(ref_pdf) > 1e-6 * np.percentile(np.abs(ref_pdf), 99)
        e = rel(pdf, ref_pdf)[ok]
        row = {"sample_pdf_p99": float(np.percentile(e, 99)), "sample_pdf_max": float(e.max()), "wo_max": float(np.abs(wo - ref_wo).max())}
        orc = O.Oracle(fw)
        for wi3, wo3, key in ((p["pdf_wi3"], p["pdf_wo3"], "pdf"), (p["wi3"], p["sample_wo3"], "pdf_of_samples")):
            got = s.plugin_pdf(t(wi3), t(wo3), T=T, variant=variant).cpu().numpy().astype(np.float64)
            want = O.plugin_pdf_disk(orc, wi3, wo3, T=T) if fw.domain == 0 else \
                O.plugin_pdf_spherical(orc, wi3.astype(np.float64), wo3.astype(np.float64), T=T, full_sphere=full)
            okp = np.abs(want) > 1e-6 * np.percentile(np.abs(want), 99)
            ee = rel(got, want)[okp]
            row[key + "_p99"] = float(np.percentile(ee, 99))
            row[key + "_max"] = float(ee.max())
        out[stem] = row
        s.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
