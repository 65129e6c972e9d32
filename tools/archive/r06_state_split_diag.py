# CPU: what does the 32-query kernels' layer 1 cost in accuracy?  Its operands (theta, sin phi, cos phi, alpha / x0, x1, alpha) enter
# one fp16 MFMA as hi + lo (hi = 11 bits by truncation, lo = fp16(x - hi)); the 16-query kernels feed them as fp32.  fp64 oracle with
# only that quantisation of the layer-1 state operands (and, separately, of the layer-1 state columns of W1).
import sys, os, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np
import parity77 as P
from oracle import bsdf_oracle as O
LOG2E = 1.0 / np.log(2.0)
RN = os.environ.get("RN") == "1"
def q22(x):
    x32 = np.asarray(x, np.float64).astype(np.float32)
    hi = x32.astype(np.float16).astype(np.float32) if RN else (x32.view(np.uint32) & np.uint32(0xFFFFE000)).view(np.float32)
    lo = (x32 - hi).astype(np.float16).astype(np.float32)
    return hi.astype(np.float64) + lo.astype(np.float64)
def rep2(w, pre=1.0):
    ws = (np.asarray(w, np.float64) * pre).astype(np.float32)
    hi = ws.astype(np.float16).astype(np.float32)
    lo = (ws - hi).astype(np.float16).astype(np.float32)
    return (hi.astype(np.float64) + lo.astype(np.float64)) / pre
class Q(O.Oracle):
    quant = True
    def velocity_jacobian(self, x, alpha, pe_cond):
        # the kernel quantises the net INPUT built from the state; the state itself stays fp32
        if self.domain == O.DOMAIN_DISK:
            return super().velocity_jacobian(x, alpha, pe_cond) if not self.quant else self._vj(x, alpha, pe_cond)
        return self._vj(x, alpha, pe_cond)
    def _vj(self, x, alpha, pe_cond):
        n = x.shape[0]
        one, zero = np.ones((n, 1)), np.zeros((n, 1))
        if self.domain == O.DOMAIN_DISK:
            s, t0, t1 = q22(x), np.concatenate([one, zero], 1), np.concatenate([zero, one], 1)
        else:
            sp, cp = np.sin(x[:, 1:2]), np.cos(x[:, 1:2])
            s = q22(np.concatenate([x[:, 0:1], sp, cp], 1))
            t0 = np.concatenate([one, zero, zero], 1)
            t1 = np.concatenate([zero, s[:, 2:3], -s[:, 1:2]], 1)
        sd = self.state_dim
        a = np.full((n, 1), float(q22(np.array([alpha]))[0]))
        h = np.concatenate([s, a, pe_cond], 1)
        w1 = self.w_in
        z = h @ w1.T
        zt0, zt1 = t0 @ w1[:, :sd].T, t1 @ w1[:, :sd].T
        h, g = O._silu_and_grad(z)
        t0, t1 = zt0 * g, zt1 * g
        for w in self.w_hidden:
            z, zt0, zt1 = h @ w.T, t0 @ w.T, t1 @ w.T
            h, g = O._silu_and_grad(z)
            t0, t1 = zt0 * g, zt1 * g
        return h @ self.w_out.T, t0 @ self.w_out.T, t1 @ self.w_out.T
n = 16384
for stem, dom, full in (("bsdf_24_spherical", "spherical", True), ("chm_orange_rgb_spherical", "spherical", False), ("bsdf_18_spherical", "spherical", True),
                        ("cc_amber_citrine_rgb_disk", "disk", False)):
    inp = P.make_inputs(stem, dom, full, n)
    fw = P._load(stem, dom)
    wi, wl, x0 = (inp[k].astype(np.float64) for k in ("wi3", "wl3", "x0"))
    T = 4 if dom == "disk" else 8
    def run(orc):
        with np.errstate(all="ignore"):
            if dom == "disk":
                _, ps, acc = O.plugin_sample_disk(orc, wi, x0, T=T, return_acc=True)
                pb, accb = O.plugin_pdf_disk(orc, wi, wl, T=T, return_acc=True)
            else:
                _, ps, acc = O.plugin_sample_spherical(orc, wi, x0, T=T, full_sphere=full, return_acc=True)
                pb, accb = O.plugin_pdf_spherical(orc, wi, wl, T=T, full_sphere=full, return_acc=True)
        return (ps, acc), (pb, accb)
    want = run(O.Oracle(fw, np.float64))
    sd = 2 if dom == "disk" else 3
    f2 = copy.copy(fw); w = np.array(fw.w_in, np.float64); w[:, :sd + 1] = rep2(w[:, :sd + 1], -LOG2E); f2.w_in = w
    f3 = copy.copy(fw); f3.base_w1 = rep2(fw.base_w1)
    for name, orc in (("state operands of layer 1 as hi + lo (22 bits)", Q(fw, np.float64)), ("W1 state / alpha columns two-part", O.Oracle(f2, np.float64)),
                      ("both", Q(f2, np.float64)), ("base net W1 two-part", O.Oracle(f3, np.float64))):
        got = run(orc)
        out = []
        for (g, _), (wv, acc) in zip(got, want):
            sc = np.percentile(np.abs(wv[np.isfinite(wv)]), 99)
            ok = np.isfinite(wv) & (np.abs(wv) > 1e-6 * sc) & (np.abs(acc) > 1e-3) & (np.abs(acc) < 1e3)
            e = np.abs(g - wv)[ok] / np.abs(wv)[ok]
            out.append(f"p50 {np.median(e):.1e} p99 {np.percentile(e, 99):.1e}")
        print(f"{stem:28s} {name:48s} sample: {out[0]} | pdf fresh: {out[1]}")
