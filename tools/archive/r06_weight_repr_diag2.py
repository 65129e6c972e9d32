# CPU: which two-part fp16 weight REPRESENTATIONS (fixed perturbations of the model) cost how much pdf() accuracy, disk nets.
# fp64 oracle arithmetic throughout; only the stored matrices are replaced by what the kernels' images hold:
#   two-part fp16 (hi = RN, lo = RN(w - hi): lo is SUBNORMAL in fp16 when |w| < 2^-2, absolute error 2^-25) as shipped, or the same
#   with the matrix scaled by a power of two first so that lo stays normal (what a scaled fold would hold).
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np
import parity77 as P
from oracle import bsdf_oracle as O

def rep2(w, pre=1.0):
    ws = np.asarray(w, np.float64) * pre
    w32 = ws.astype(np.float32)
    hi = w32.astype(np.float16).astype(np.float32)
    lo = (w32 - hi).astype(np.float16).astype(np.float32)
    return (hi.astype(np.float64) + lo.astype(np.float64)) / pre

def pow2_scale(w, top=2.0 ** 13):
    m = np.abs(w).max()
    return 2.0 ** np.floor(np.log2(top / m)) if m > 0 else 1.0

class MimOracle(O.Oracle):
    """Disk 25-32x3-2 net, Jacobian by meeting in the middle with separately represented matrices."""
    def __init__(self, fw, mode):
        super().__init__(fw, np.float64)
        W1, (W2, W3), Wo = self.w_in, self.w_hidden, self.w_out
        self.mode = mode
        f = {"exact": lambda w: w, "rep": rep2, "scaled": lambda w: rep2(w, pow2_scale(w))}
        fold = f[mode.get("fold", "exact")]
        hid = f[mode.get("hidden", "exact")]
        out = f[mode.get("out", "exact")]
        self.F = [fold(W2 * W1[:, i][None, :]) for i in range(2)]          # F_i = W2 diag(W1[:, i])
        self.G = [fold(W3.T * Wo[j][None, :]) for j in range(2)]           # G_j = W3^T diag(Wout[j])
        self.W2f, self.W3f, self.Wof = hid(W2), hid(W3), out(Wo)
        self.W1f = f[mode.get("in", "exact")](W1)
    def velocity_jacobian(self, x, alpha, pe_cond):
        n = x.shape[0]
        h0 = np.concatenate([x, np.full((n, 1), alpha), pe_cond], 1)
        z1 = h0 @ self.W1f.T
        h1, g1 = O._silu_and_grad(z1)
        z2 = h1 @ self.W2f.T
        h2, g2 = O._silu_and_grad(z2)
        z3 = h2 @ self.W3f.T
        h3, g3 = O._silu_and_grad(z3)
        v = h3 @ self.Wof.T
        U = [g1 @ Fi.T for Fi in self.F]
        R = [g3 @ Gj.T for Gj in self.G]
        J = lambda j, i: (R[j] * g2 * U[i]).sum(1)
        d0 = np.stack([J(0, 0), J(1, 0)], 1)
        d1 = np.stack([J(0, 1), J(1, 1)], 1)
        return v, d0, d1

def main():
    sets = sys.argv[1:] or ["cc_amber_citrine_rgb_disk", "chm_light_blue_rgb_disk", "ilm_solo_m_68_rgb_disk", "aniso_miro_7_rgb_disk", "chm_orange_rgb_disk"]
    n = 16384
    modes = {"all exact (check)": {}, "Wout rep (16-query kernels)": {"out": "rep"}, "Wout rep, lo scaled": {"out": "scaled"},
             "folds F, G rep (both families)": {"fold": "rep"}, "folds rep, scaled": {"fold": "scaled"},
             "hidden W2, W3 rep (both)": {"hidden": "rep"}, "hidden rep, scaled (not implementable as is)": {"hidden": "scaled"},
             "W1 rep (32-query)": {"in": "rep"},
             "as the 32-query kernels": {"fold": "rep", "hidden": "rep", "in": "rep"},
             "as the 16-query kernels": {"fold": "rep", "hidden": "rep", "out": "rep"},
             "32-query with scaled folds": {"fold": "scaled", "hidden": "rep", "in": "rep"},
             "16-query with scaled folds and Wout lo": {"fold": "scaled", "hidden": "rep", "out": "scaled"}}
    for stem in sets:
        inp = P.make_inputs(stem, "disk", False, n)
        fw = P._load(stem, "disk")
        wi, wl, x0 = (inp[k].astype(np.float64) for k in ("wi3", "wl3", "x0"))
        base = O.Oracle(fw, np.float64)
        with np.errstate(all="ignore"):
            want_b, acc_b = O.plugin_pdf_disk(base, wi, wl, T=4, return_acc=True)
            _, want_s, acc_s = O.plugin_sample_disk(base, wi, x0, T=4, return_acc=True)
        print(f"== {stem}: |W_out| median {np.median(np.abs(fw.w_out)):.3g} min {np.abs(fw.w_out).min():.3g}; |F| median "
              f"{np.median(np.abs(fw.w_hidden[0] * fw.w_in[:, 0][None, :])):.3g}; |W2| median {np.median(np.abs(fw.w_hidden[0])):.3g}")
        for name, mode in modes.items():
            orc = MimOracle(fw, mode)
            with np.errstate(all="ignore"):
                got_b, _ = O.plugin_pdf_disk(orc, wi, wl, T=4, return_acc=True)
                _, got_s, _ = O.plugin_sample_disk(orc, wi, x0, T=4, return_acc=True)
            out = []
            for got, want, acc in ((got_s, want_s, acc_s), (got_b, want_b, acc_b)):
                sc = np.percentile(np.abs(want), 99)
                ok = (np.abs(want) > 1e-6 * sc) & (np.abs(acc) > 1e-3) & (np.abs(acc) < 1e3)
                e = np.abs(got - want)[ok] / np.abs(want)[ok]
                out.append(f"p50 {np.median(e):.1e} p99 {np.percentile(e, 99):.1e} mean {((got - want)[ok] / np.abs(want)[ok]).mean():+.1e}")
            print(f"  {name:46s} sample: {out[0]} | pdf fresh: {out[1]}")

main()
