# the one row of 262 144 (bsdf_5_spherical, sample) whose density has the other sign than the fp64 oracle's: what is it?
import sys, os, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np
import parity77 as P
stem, dom, full = "bsdf_5_spherical", "spherical", True
inp = P.make_inputs(stem, dom, full, 262144)
g = P.gpu_eval(stem, dom, full, inp)
_, o = P.oracle_eval((stem, dom, full, inp, g["wo_a"]))
want, want32, acc = o["f64"]["sample"], o["f32"]["sample"].astype(np.float64), o["f64"]["sample_acc"]
for tile in (32, 16):
    got = g[tile]["sample"].astype(np.float64)
    scale = np.percentile(np.abs(want), 99)
    ok = np.abs(want) > 1e-6 * scale
    bad = np.where(ok & (np.sign(got) != np.sign(want)))[0]
    for i in bad:
        print(json.dumps({"tile": tile, "row": int(i), "kernel": got[i], "oracle_fp64": want[i], "oracle_fp32": want32[i], "prod_det": acc[i],
                          "p99_scale": scale, "wi": inp["wi3"][i].tolist(), "x0": inp["x0"][i].tolist()}))
