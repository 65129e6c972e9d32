# CPU: how much pdf() error does the fp16 hi+lo REPRESENTATION of single weight matrices cause on cc_amber_citrine_rgb_disk at fresh
# directions (fp64 oracle with one matrix replaced by its two-part fp16 form)?  A fixed perturbation of the model, not rounding noise.
import sys, os, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np
import parity77 as P
from oracle import bsdf_oracle as O
stem = sys.argv[1] if len(sys.argv) > 1 else "cc_amber_citrine_rgb_disk"
n = 32768
inp = P.make_inputs(stem, "disk", False, n)
fw = P._load(stem, "disk")
def split2(w, scale=1.0):
    ws = (w.astype(np.float64) * scale).astype(np.float32)
    hi = ws.astype(np.float16).astype(np.float32)
    lo = (ws - hi).astype(np.float16).astype(np.float32)
    return ((hi.astype(np.float64) + lo.astype(np.float64)) / scale)
def run(fw_):
    orc = O.Oracle(fw_, np.float64)
    with np.errstate(all="ignore"):
        return O.plugin_pdf_disk(orc, inp["wi3"].astype(np.float64), inp["wl3"].astype(np.float64), T=4, return_acc=True)
want, acc = run(fw)
scale = np.percentile(np.abs(want), 99)
ok = (np.abs(want) > 1e-6 * scale) & (np.abs(acc) > 1e-3) & (np.abs(acc) < 1e3)
print("fields:", [k for k in vars(fw) if k.startswith(("w_", "base_"))])
def report(name, fw_):
    got, _ = run(fw_)
    e = np.abs(got - want)[ok] / np.abs(want)[ok]
    s = ((got - want)[ok] / np.abs(want)[ok]).mean()
    print(f"{name:44s} p50 {np.median(e):.2e} p99 {np.percentile(e, 99):.2e} signed mean {s:+.2e}")
LN2 = float(np.log(2.0)); LOG2E = 1.0 / LN2
for name, field, sc in (("w_out as hi+lo of (-ln2 w)", "w_out", -LN2), ("w_out rounded to fp32 after scaling only", "w_out", None),
                        ("w_hidden as hi+lo", "w_hidden", 1.0), ("w_in as hi+lo of (-log2e w)", "w_in", -LOG2E),
                        ("w_in rounded to fp32 after scaling only", "w_in", None)):
    f2 = copy.copy(fw)
    w = getattr(fw, field)
    if sc is None:
        s_ = -LN2 if field == "w_out" else -LOG2E
        setattr(f2, field, ((w.astype(np.float64) * s_).astype(np.float32).astype(np.float64) / s_))
    else:
        setattr(f2, field, split2(w, sc))
    report(name, f2)
