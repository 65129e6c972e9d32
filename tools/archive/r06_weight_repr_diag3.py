# CPU: the same question for the spherical / full-sphere sets the 32-query kernels read their worst figures on — one matrix at a time
# replaced by its two-part fp16 form (fp64 oracle arithmetic otherwise).
import sys, os, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np
import parity77 as P
from oracle import bsdf_oracle as O
LN2 = float(np.log(2.0)); LOG2E = 1.0 / LN2
def rep2(w, pre=1.0, lo_scale=1.0):
    ws = (np.asarray(w, np.float64) * pre).astype(np.float32)
    hi = ws.astype(np.float16).astype(np.float32)
    lo = ((ws - hi) * np.float32(lo_scale)).astype(np.float16).astype(np.float32)
    return (hi.astype(np.float64) + lo.astype(np.float64) / lo_scale) / pre
n = 16384
for stem, dom, full in (("bsdf_24_spherical", "spherical", True), ("chm_orange_rgb_spherical", "spherical", False), ("bsdf_18_spherical", "spherical", True)):
    inp = P.make_inputs(stem, dom, full, n)
    fw = P._load(stem, dom)
    wi, wl, x0 = (inp[k].astype(np.float64) for k in ("wi3", "wl3", "x0"))
    def run(fw_):
        orc = O.Oracle(fw_, np.float64)
        with np.errstate(all="ignore"):
            wo, ps, acc = O.plugin_sample_spherical(orc, wi, x0, T=8, full_sphere=full, return_acc=True)
            pb, accb = O.plugin_pdf_spherical(orc, wi, wl, T=8, full_sphere=full, return_acc=True)
        return (ps, acc), (pb, accb)
    want = run(fw)
    print(f"== {stem}: |w_in[:, :4]| median {np.median(np.abs(fw.w_in[:, :4])):.3g}, |base_w1| median {np.median(np.abs(fw.base_w1)):.3g}, |base_w2| {np.median(np.abs(fw.base_w2)):.3g}")
    for name, field, kw in (("w_in two-part (x -log2e)", "w_in", dict(pre=-LOG2E)), ("w_in two-part, lo x 2^11", "w_in", dict(pre=-LOG2E, lo_scale=2048.0)),
                            ("w_hidden two-part", "w_hidden", {}), ("w_hidden two-part, lo x 2^11", "w_hidden", dict(lo_scale=2048.0)),
                            ("base_w1 two-part", "base_w1", {}), ("base_w1 two-part, lo x 2^11", "base_w1", dict(lo_scale=2048.0))):
        f2 = copy.copy(fw)
        w = getattr(fw, field)
        setattr(f2, field, [rep2(x, **kw) for x in w] if isinstance(w, (list, tuple)) else rep2(w, **kw))
        got = run(f2)
        out = []
        for (g, _), (wv, acc) in zip(got, want):
            sc = np.percentile(np.abs(wv[np.isfinite(wv)]), 99)
            ok = np.isfinite(wv) & (np.abs(wv) > 1e-6 * sc) & (np.abs(acc) > 1e-3) & (np.abs(acc) < 1e3)
            e = np.abs(g - wv)[ok] / np.abs(wv)[ok]
            out.append(f"p50 {np.median(e):.1e} p99 {np.percentile(e, 99):.1e}")
        print(f"  {name:34s} sample: {out[0]} | pdf fresh: {out[1]}")
