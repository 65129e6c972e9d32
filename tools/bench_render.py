"""Config 5 (SURVEY.md §8 d / f3): matpreview-style render, 512^2 x (passes x 4 spp), one material ball.
Reports whole-pass throughput and the split between the harness kernels and the hot path.
  python tools/bench_render.py [--plugin disk|spherical] [--passes 64] [--spp 4] [--size 512]
N>1: launch with torch.distributed.run (image rows split over ranks, final film gather)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--plugin", default="disk", choices=["disk", "spherical"])
ap.add_argument("--material", default="aniso_miro_7_rgb")
ap.add_argument("--passes", type=int, default=64)
ap.add_argument("--spp", type=int, default=4)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--out", default=None, help="write the image as .npy")
ap.add_argument("--measured-dir", default=None, help="directory with <material>.bsdf: shade with the ground-truth f")
a = ap.parse_args()
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
torch.cuda.set_device(local)
if world > 1:
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(os.environ.get("BSDFD_BENCH_BACKEND", "nccl"))
from bsdf_diffusion_sampling_amd import wavefront as WF
from bsdf_diffusion_sampling_amd.sharding import shard_range
if a.plugin == "disk":
    from bsdf_diffusion_sampling_amd.brdf_measured_disk import MyBSDF
else:
    from bsdf_diffusion_sampling_amd.brdf_measured_spherical import MyBSDF
props = {"filename": a.material, "albedo": [0.9, 0.9, 0.9]}
if a.measured_dir:
    props["measured_dir"] = a.measured_dir
else:
    props["measured"] = False
plug = MyBSDF(props)
r = WF.WavefrontRenderer(plug, WF.Camera(width=a.size, height=a.size))
r0, r1 = shard_range(a.size, rank, world)
t0 = time.time()
while time.time() - t0 < 0.2:           # leave the idle clocks (tools/ramp.py)
    r.render(1, a.spp, seed=99, rows=(r0, r1)); torch.cuda.synchronize()
if world > 1: dist.barrier()
torch.cuda.synchronize(); t0 = time.perf_counter()
img = r.render_sharded(a.passes, a.spp, seed=0)
torch.cuda.synchronize()
if world > 1: dist.barrier()
dt = time.perf_counter() - t0
# split of one pass on this rank's tile (events on the current stream)
n = (r1 - r0) * a.size * a.spp
b = r._buffers(n); film = torch.zeros((r1 - r0, a.size, 3), device=r.device)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
acc = [0.0] * 5
for k in range(10):
    ev[0].record(); r.primary(r0, r1, a.spp, 0, k)
    ev[1].record(); plug.sampler.plugin_sample(b["wi"], None, T=plug.T, variant=plug.VARIANT, seed=k, offset=0, out=(b["wo"], b["pdf_o"]))
    ev[2].record(); plug.sampler.plugin_pdf(b["wi"], b["wl"], T=plug.T, variant=plug.VARIANT, out=b["pdf_l"])
    ev[3].record()
    if r.use_ground_truth:
        plug.bsdf.eval_t(b["wi"], b["wo"], out=b["f_o"]); plug.bsdf.eval_t(b["wi"], b["wl"], out=b["f_l"])
    ev[4].record(); r.shade(r0, r1, a.spp, b, film)
    ev[5].record(); torch.cuda.synchronize()
    for i in range(5): acc[i] += ev[i].elapsed_time(ev[i + 1]) / 10
if rank == 0:
    paths = a.size * a.size * a.spp * a.passes
    print(json.dumps({"workload": f"render_{a.size}x{a.size}_{a.passes}x{a.spp}spp_{a.plugin}", "n_gpus": world,
                      "material": a.material, "euler_steps": plug.T, "seconds": dt, "passes_per_s": a.passes / dt,
                      "Mpaths_per_s": paths / dt / 1e6, "sampler_calls_per_path": 2,
                      "ground_truth_eval": bool(r.use_ground_truth),
                      "ms_per_pass_split": {"primary": acc[0], "sample": acc[1], "pdf": acc[2], "eval_x2": acc[3], "shade": acc[4]},
                      "hot_path_fraction": (acc[1] + acc[2]) / sum(acc),
                      "image_mean": float(img.mean()), "image_finite": bool(torch.isfinite(img).all())}))
    if a.out: 
        import numpy as np; np.save(a.out, img.cpu().numpy())
if world > 1: dist.destroy_process_group()
