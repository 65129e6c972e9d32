import sys, os, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import load_case
from oracle import bsdf_oracle as O
from bsdf_diffusion_sampling_amd.sampler import FlowSampler
import bench
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
g, fw = load_case("aniso_miro_7_rgb_spherical_complex")
n, T = 1024, 128
wi, x0 = g["wi"][:n], g["x0"][:n]
xo, _ = O.Oracle(fw).flow(x0, wi, T, reverse=False)
out = {}
for tile in (16, 32):
    s = FlowSampler(fw, precision="f16", tile=tile)
    x = s.flow_samples_only(t(wi), t(x0), T=T).cpu().numpy()
    e = np.abs(x - xo).max(1)
    out[f"t{tile}"] = {"p50": float(np.percentile(e, 50)), "p99": float(np.percentile(e, 99)), "max": float(e.max()), "nan": int(np.isnan(x).sum())}
    # ragged
    xr = s.flow_samples_only(t(wi[:333]), t(x0[:333]), T=T).cpu().numpy()
    out[f"t{tile}"]["ragged_equal"] = bool(np.array_equal(xr, x[:333]))
    s.close()
print(json.dumps(out))
# timing: bench Teacher workload, alternating
res = {16: [], 32: []}
for rnd in range(3):
    for tile in (16, 32):
        os.environ["BSDFD_TILE"] = str(tile)
        wl = bench.Teacher("teacher_64x6_4Mi_T128", torch.device("cuda"), 0, "f16")
        for _ in range(2): wl.run_pass(0)
        torch.cuda.synchronize(); wl.smp.set_profiling(True)
        for _ in range(4): wl.run_pass(0)
        k, ms = wl.smp.profile_read(); mhz = wl.smp.profile_clock_mhz(); wl.smp.set_profiling(False)
        res[tile].append((ms / k, mhz))
        del wl; torch.cuda.empty_cache()
for tile in (16, 32):
    print(tile, [(round(a, 3), round(b)) for a, b in res[tile]])
print("ratio", np.median([a for a, _ in res[32]]) / np.median([a for a, _ in res[16]]))
