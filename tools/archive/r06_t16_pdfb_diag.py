# cc_amber_citrine_rgb_disk, pdf() at fresh directions: the 16-query kernels read 1.05e-4 where the 32-query ones (and the exact-fp32
# 16-query ones) read 3.3e-5.  Which rows?  (relative error by radius of the asked direction, by |prod det J|, by the density)
import sys, os, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np
import parity77 as P
stem, dom, full = sys.argv[1] if len(sys.argv) > 1 else "cc_amber_citrine_rgb_disk", "disk", False
n = 65536
inp = P.make_inputs(stem, dom, full, n)
g = P.gpu_eval(stem, dom, full, inp)
g32f = P.gpu_eval(stem, dom, full, inp, tiles=(16,), precision="f32")
_, o = P.oracle_eval((stem, dom, full, inp, g["wo_a"]))
want, want32, acc = o["f64"]["pdf_b"], o["f32"]["pdf_b"].astype(np.float64), o["f64"]["pdf_b_acc"]
scale = np.percentile(np.abs(want), 99)
ok = (np.abs(want) > 1e-6 * scale) & (np.abs(acc) > 1e-3) & (np.abs(acc) < 1e3)
r = np.hypot(inp["wl3"][:, 0], inp["wl3"][:, 1]).astype(np.float64)
ri = np.hypot(inp["wi3"][:, 0], inp["wi3"][:, 1]).astype(np.float64)
E = {"t32": g[32]["pdf_b"], "t16": g[16]["pdf_b"], "t16_f32": g32f[16]["pdf_b"], "ref32": want32}
E = {k: np.abs(v.astype(np.float64) - want) / np.abs(want) for k, v in E.items()}
print("rows", int(ok.sum()), {k: float(np.percentile(v[ok], 99)) for k, v in E.items()})
def table(name, x, edges):
    print("by", name)
    for a, b in zip(edges[:-1], edges[1:]):
        m = ok & (x >= a) & (x < b)
        if m.sum() < 50: continue
        print(f"  [{a:.4g}, {b:.4g}) rows {int(m.sum()):6d} " + " ".join(f"{k} p50 {np.median(v[m]):.2e} p99 {np.percentile(v[m], 99):.2e}" for k, v in E.items()))
table("r_o", r, [0, .3, .6, .8, .9, .95, .98, .99, .995, 1.0001])
table("r_i", ri, [0, .3, .6, .8, .9, .95, 1.0])
table("|prod det|", np.abs(acc), [1e-3, 1e-2, 1e-1, 0.5, 2, 10, 100, 1e3])
table("pdf / p99", np.abs(want) / scale, [1e-6, 1e-4, 1e-2, 1e-1, 1, 1e9])
# signed error: is the 16-query error a bias?
s16 = (g[16]["pdf_b"].astype(np.float64) - want) / np.abs(want)
s32 = (g[32]["pdf_b"].astype(np.float64) - want) / np.abs(want)
print("signed mean t16 %.3e t32 %.3e ; corr(t16, t32) %.3f" % (s16[ok].mean(), s32[ok].mean(), np.corrcoef(s16[ok], s32[ok])[0, 1]))
top = np.argsort(-np.where(ok, E["t16"], 0))[:12]
for i in top:
    print(json.dumps({"row": int(i), "t16": E["t16"][i], "t32": E["t32"][i], "t16_f32": E["t16_f32"][i], "ref32": E["ref32"][i], "r_o": r[i], "r_i": ri[i], "acc": acc[i], "pdf": want[i] / scale}))
