# CPU: the disk kernels' conditioning term c = W1[:, PE] PE(omega_i) runs on split-fp16 MFMAs: PE values as hi + lo with a TRUNCATED hi
# (22 bits).  What does that quantisation cost, and what would a rounded hi (23 bits) give?  fp64 oracle, only the PE values quantised.
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np
import parity77 as P
from oracle import bsdf_oracle as O
def q(x, rn):
    x32 = np.asarray(x, np.float64).astype(np.float32)
    hi = x32.astype(np.float16).astype(np.float32) if rn else (x32.view(np.uint32) & np.uint32(0xFFFFE000)).view(np.float32)
    lo = (x32 - hi).astype(np.float16).astype(np.float32)
    return hi.astype(np.float64) + lo.astype(np.float64)
class Q(O.Oracle):
    rn = False
    def velocity_jacobian(self, x, alpha, pe_cond):
        return super().velocity_jacobian(x, alpha, q(pe_cond, self.rn))
n = 16384
for stem in sys.argv[1:] or ["cc_amber_citrine_rgb_disk", "ilm_solo_m_68_rgb_disk", "chm_light_blue_rgb_disk", "aniso_miro_7_rgb_disk"]:
    inp = P.make_inputs(stem, "disk", False, n)
    fw = P._load(stem, "disk")
    wi, wl, x0 = (inp[k].astype(np.float64) for k in ("wi3", "wl3", "x0"))
    def run(orc):
        with np.errstate(all="ignore"):
            _, ps, acc = O.plugin_sample_disk(orc, wi, x0, T=4, return_acc=True)
            pb, accb = O.plugin_pdf_disk(orc, wi, wl, T=4, return_acc=True)
        return (ps, acc), (pb, accb)
    want = run(O.Oracle(fw, np.float64))
    for rn in (False, True):
        orc = Q(fw, np.float64); orc.rn = rn
        got = run(orc)
        out = []
        for (g, _), (wv, acc) in zip(got, want):
            sc = np.percentile(np.abs(wv), 99)
            ok = (np.abs(wv) > 1e-6 * sc) & (np.abs(acc) > 1e-3) & (np.abs(acc) < 1e3)
            e = np.abs(g - wv)[ok] / np.abs(wv)[ok]
            out.append(f"p50 {np.median(e):.1e} p99 {np.percentile(e, 99):.1e}")
        print(f"{stem:28s} PE values hi + lo, hi {'rounded  ' if rn else 'truncated'}: sample {out[0]} | pdf fresh {out[1]}")
