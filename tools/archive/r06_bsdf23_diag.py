# CPU: bsdf_23_spherical, pdf() at fresh directions — the one exempt (set, call): kernels 4.9e-4, the reference's fp32 1.4e-3.  What dominates?
import sys, os, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np
import parity77 as P
from oracle import bsdf_oracle as O
stem, dom, full = "bsdf_23_spherical", "spherical", True
n = 32768
inp = P.make_inputs(stem, dom, full, n)
fw = P._load(stem, dom)
wi, wl = inp["wi3"].astype(np.float64), inp["wl3"].astype(np.float64)
def run(orc, wi_=wi, wl_=wl):
    with np.errstate(all="ignore"):
        return O.plugin_pdf_spherical(orc, wi_, wl_, T=8, full_sphere=full, return_acc=True)
want, acc = run(O.Oracle(fw, np.float64))
sc = np.percentile(np.abs(want[np.isfinite(want)]), 99)
ok = np.isfinite(want) & (np.abs(want) > 1e-6 * sc) & (np.abs(acc) > 1e-3) & (np.abs(acc) < 1e3)
def rep(name, got):
    e = np.abs(got - want)[ok] / np.abs(want)[ok]
    print(f"{name:70s} rows {int(ok.sum())} p50 {np.median(e):.1e} p99 {np.percentile(e, 99):.1e}")
got32, _ = run(O.Oracle(fw, np.float32), wi.astype(np.float32), wl.astype(np.float32))
rep("the reference's arithmetic in fp32 (acos as written)", got32.astype(np.float64))
# inputs perturbed by one fp32 ulp-class relative error: how ill-conditioned is the map itself?
rng = np.random.default_rng(1)
for eps in (2.0 ** -24, 2.0 ** -22):
    wlp = wl * (1 + eps * rng.standard_normal(wl.shape))
    g, _ = run(O.Oracle(fw, np.float64), wi, wlp)
    rep(f"fp64 arithmetic, omega_o components perturbed by {eps:.1e} relative", g)
# where the rows are
th = np.arccos(np.clip(wl[:, 2], -1, 1))
e32 = np.abs(got32 - want) / np.abs(want)
for a, b in ((0, 0.05), (0.05, 0.2), (0.2, 1.4), (1.4, 1.75), (1.75, 2.9), (2.9, 3.1), (3.1, 3.15)):
    m = ok & (th >= a) & (th < b)
    if m.sum() > 30:
        print(f"  theta_o in [{a}, {b}): rows {int(m.sum()):6d} reference-fp32 p50 {np.median(e32[m]):.1e} p99 {np.percentile(e32[m], 99):.1e}")
