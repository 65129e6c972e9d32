"""Render the 12-ball array scene (the shape of matpreview/disney_bsdf_array0_envmap.xml) with a MaterialTable."""
import sys, os, time, json, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bsdf_diffusion_sampling_amd import wavefront as WF, weights as W
from bsdf_diffusion_sampling_amd.materials import MaterialTable
from bsdf_diffusion_sampling_amd.render_cli import write_png, tonemap
ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=683); ap.add_argument("--height", type=int, default=512)
ap.add_argument("--passes", type=int, default=64); ap.add_argument("--spp", type=int, default=4)
ap.add_argument("--domain", default="disk"); ap.add_argument("--measured-dir", default=None)
ap.add_argument("--out", default="gpurun_out/array0")
a = ap.parse_args()
cam, centers, radii = WF.array0_scene(a.width, a.height)
stems = [m + "_" + a.domain for m in WF.ARRAY0_MATERIALS]
tab = MaterialTable(stems)
gts = {}
if a.measured_dir:
    from bsdf_diffusion_sampling_amd.measured import MeasuredBSDF, find_measured_file
    for i, m in enumerate(WF.ARRAY0_MATERIALS):
        p = find_measured_file(m, a.measured_dir)
        if p: gts[i] = MeasuredBSDF(p)
r = WF.ArrayRenderer(tab, centers, radii, camera=cam, ground_truth=gts)
r.render(2, a.spp, seed=9); torch.cuda.synchronize()
t0 = time.perf_counter(); img = r.render(a.passes, a.spp, seed=0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
b = r.primary(0, a.height, 1, 0, 0); mat = b["mat"].cpu().numpy()
paths = a.width * a.height * a.spp * a.passes
print(json.dumps({"workload": f"array0_{a.width}x{a.height}_{a.passes}x{a.spp}spp_{a.domain}", "materials": len(tab),
                  "ground_truth_materials": len(gts), "seconds": dt, "passes_per_s": a.passes / dt, "Mpaths_per_s": paths / dt / 1e6,
                  "ball_fraction": float((mat < 12).mean()), "floor_fraction": float((mat == 12).mean()), "miss_fraction": float((mat == 13).mean())}))
img = img.cpu().numpy(); os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
np.save(a.out + ".npy", img); write_png(a.out + ".png", tonemap(img))
