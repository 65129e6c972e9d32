#!/usr/bin/env python3
"""One measurement pass of ONE build of libbsdfd.so (selected with BSDFD_LIB_PATH): accuracy on the golden
cases and kernel time (HIP events on the launch stream, median over reps) of the tracked workloads.
Prints one JSON line.  `tools/ab_run.sh` interleaves several builds over several rounds (cdna guide §5.4 rule 24:
perf deltas from interleaved rounds, not from separate one-shot runs).

    BSDFD_LIB_PATH=/path/to/variant.so python tools/ab.py [--tag NAME] [--only disk8,disk4,...] [--acc]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from bsdf_diffusion_sampling_amd import _lib  # noqa: E402
from bsdf_diffusion_sampling_amd import weights as W  # noqa: E402
from bsdf_diffusion_sampling_amd.sampler import FlowSampler  # noqa: E402


LAST_MHZ = [None]   # in-kernel shader clock of the launches kernel_ms() timed last (bsdfd_profile_clock_mhz)


def kernel_ms(smp, fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    out, mhz = [], []
    for _ in range(reps):
        smp.set_profiling(True)
        fn()
        n, ms = smp.profile_read()
        out.append(ms / max(n, 1))
        mhz.append(smp.profile_clock_mhz())
    smp.set_profiling(False)
    LAST_MHZ[0] = float(np.median(mhz)) if mhz else None
    return float(np.median(out)), float(np.min(out))


def energy(run, n, seconds):
    """J per million queries + socket watts of `run` back to back (bsdf_diffusion_sampling_amd/power.py): the flow kernels are
    power-limited, so an A/B is decided by joules per query — time follows."""
    if seconds <= 0:
        return {}
    from bsdf_diffusion_sampling_amd.power import energy_probe
    e = energy_probe(lambda k: run(), n, seconds=seconds, sync=torch.cuda.synchronize)
    return {"joule_per_Mquery": e["joule_per_Mquery"], "watts": e["socket_power_w"], "sclk_mhz": e["sclk_mhz"], "power_samples": e["samples"]}


def settle(smp, wi, T, ms=150.0):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        smp.plugin_sample(wi, None, T=T)
        torch.cuda.synchronize()


def accuracy():
    from conftest import GOLDEN_CASES, load_case
    from oracle import bsdf_oracle as O
    dev = torch.device("cuda")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
    res = {}
    for stem in GOLDEN_CASES:
        g, fw = load_case(stem)
        T = int(g["meta_T"])
        orc = O.Oracle(fw)
        xo, po = orc.network_sampling(g["wi"], g["x0"], T)
        _, acc = orc.flow(g["x0"], g["wi"], T, False)
        ok = (np.abs(acc) > 1e-3) & (np.abs(acc) < 1e3)
        ok = ok & (np.abs(po) > 1e-6 * np.percentile(np.abs(po[ok]), 99))
        s = FlowSampler(fw, precision="split3")
        x, p = s.network_sampling(t(g["wi"]), t(g["x0"]), T=T)
        x, p = x.cpu().numpy(), p.cpu().numpy()
        r = (np.abs(p - po) / np.maximum(np.abs(po), 1e-30))[ok]
        pr = orc.network_pdf(xo, g["wi"], T)
        p2 = s.network_pdf(t(xo), t(g["wi"]), T=T).cpu().numpy()
        ok2 = np.abs(pr) > 1e-6 * np.percentile(np.abs(pr), 99)
        r2 = (np.abs(p2 - pr) / np.maximum(np.abs(pr), 1e-30))[ok & ok2]
        res[stem] = {"x_err": float(np.abs(x - xo).max()), "p99": float(np.percentile(r, 99)), "med": float(np.median(r)),
                     "pdf_p99": float(np.percentile(r2, 99)), "nan": int(np.isnan(p).sum() + np.isnan(p2).sum())}
        s.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default=os.path.basename(_lib.LIB_PATH))
    ap.add_argument("--only", default="disk8,disk4,sph8,cplx8,teacher")
    ap.add_argument("--acc", action="store_true")
    ap.add_argument("--reps", type=int, default=15)
    ap.add_argument("--energy-s", type=float, default=0.8, help="seconds of back-to-back sample()+pdf() pairs per workload for J/query (0: skip)")
    a = ap.parse_args()
    only = set(a.only.split(","))
    dev = torch.device("cuda", 0)
    out = {"tag": a.tag}
    if a.acc:
        out["acc"] = accuracy()

    def plug(name, material, domain, kind, n, T, prec="default"):
        fw = W.load(W.shipped_path(material, domain, kind) if kind else W.shipped_path(material, domain))
        smp = FlowSampler(fw, precision=prec)
        wi = bench.make_wi(domain, n, 1234, dev)
        wo = torch.empty((n, 3), device=dev)
        ps = torch.empty((n,), device=dev)
        pp = torch.empty((n,), device=dev)
        settle(smp, wi, T)
        ms_s, mn_s = kernel_ms(smp, lambda: smp.plugin_sample(wi, None, T=T, seed=3, out=(wo, ps)), a.reps)
        mhz_s = LAST_MHZ[0]
        ms_p, mn_p = kernel_ms(smp, lambda: smp.plugin_pdf(wi, wo, T=T, out=pp), a.reps)
        mhz_p = LAST_MHZ[0]
        fl = smp.flops_per_query(T) * n
        out[name] = {"sample_ms": ms_s, "pdf_ms": ms_p, "sample_min": mn_s, "pdf_min": mn_p,
                     "frac": fl / (0.5 * (ms_s + ms_p) * 1e-3) / 2.5e15, "sample_mhz": mhz_s, "pdf_mhz": mhz_p,
                     "sample_Mcycles": ms_s * (mhz_s or 0) * 1e-3, "pdf_Mcycles": ms_p * (mhz_p or 0) * 1e-3}
        out[name].update(energy(lambda: (smp.plugin_sample(wi, None, T=T, seed=3, out=(wo, ps)), smp.plugin_pdf(wi, wo, T=T, out=pp)),
                                n, a.energy_s))
        smp.close()

    if "disk8" in only:
        plug("disk8", "aniso_miro_7_rgb", "disk", None, 1 << 20, 8)
    if "disk4" in only:
        plug("disk4", "aniso_miro_7_rgb", "disk", None, 1 << 20, 4)
    if "fused4" in only:  # sample(wi) + pdf(wi, wl) in one launch (the renderer's call pattern), disk T = 4
        fw = W.load(W.shipped_path("aniso_miro_7_rgb", "disk"))
        smp = FlowSampler(fw)
        n = 1 << 20
        wi, wl = bench.make_wi("disk", n, 1234, dev), bench.make_wi("disk", n, 99, dev)
        settle(smp, wi, 4)
        ms, mn = kernel_ms(smp, lambda: smp.plugin_sample_pdf(wi, wl, None, T=4, seed=3), a.reps)
        out["fused4"] = {"ms": ms, "min": mn}
        smp.close()
    if "fusedsph8" in only:  # the same for a spherical net (T = 8): the kernel a spherical render uses
        fw = W.load(W.shipped_path("aniso_miro_7_rgb", "spherical"))
        smp = FlowSampler(fw)
        n = 1 << 20
        wi, wl = bench.make_wi("spherical", n, 1234, dev), bench.make_wi("spherical", n, 99, dev)
        settle(smp, wi, 8)
        ms, mn = kernel_ms(smp, lambda: smp.plugin_sample_pdf(wi, wl, None, T=8, seed=3), a.reps)
        out["fusedsph8"] = {"ms": ms, "min": mn}
        if a.acc:  # fused == two single-op launches (same seed): max relative difference of the two pdfs
            wo, po, pl = smp.plugin_sample_pdf(wi, wl, None, T=8, seed=3)
            wo2, po2 = smp.plugin_sample(wi, None, T=8, seed=3)
            pl2 = smp.plugin_pdf(wi, wl, T=8)
            def rel(u, v):
                m = v.abs() > 1e-6 * v.abs().median()
                return float(((u - v).abs()[m] / v.abs()[m]).quantile(0.999))
            out["fusedsph8"]["vs_two_calls_p999"] = [float((wo - wo2).abs().max()), rel(po, po2), rel(pl, pl2)]
        smp.close()
    if "sph8" in only:
        plug("sph8", "aniso_miro_7_rgb", "spherical", None, 1 << 22, 8)
    if "cplx8" in only:
        plug("cplx8", "aniso_miro_7_rgb", "spherical", "complex", 1 << 20, 8)
    if "teacher" in only:
        n, T = 1 << 22, 128
        fw = W.load(W.shipped_path("aniso_miro_7_rgb", "spherical", "complex"))
        s = FlowSampler(fw, precision="f16")
        g = torch.Generator().manual_seed(2)
        u = torch.rand(n, 2, generator=g)
        cond = torch.stack([1.5 * u[:, 0], (2 * u[:, 1] - 1) * np.pi], 1).float().to(dev)
        x0 = torch.stack([0.7 + 0.3 * torch.randn(n, generator=g), (2 * torch.rand(n, generator=g) - 1) * np.pi], 1).float().to(dev)
        ms, mn = kernel_ms(s, lambda: s.flow_samples_only(cond, x0, T=T), max(a.reps // 3, 3), warm=1)
        w = fw.width
        fwd = 2 * (fw.in_dim * w + (fw.n_hidden - 1) * w * w + 2 * w) * T
        out["teacher"] = {"ms": ms, "min": mn, "frac": n * fwd / (ms * 1e-3) / 2.5e15, "mhz": LAST_MHZ[0]}
        out["teacher"].update(energy(lambda: s.flow_samples_only(cond, x0, T=T), n, a.energy_s))
        s.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
