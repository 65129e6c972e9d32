"""Command-line render, the counterpart of the reference's ``python brdf_measured_disk.py --scene_file X``
(rendering/brdf_measured_disk.py:133-164: load the scene, 128 passes of 4 spp, write .png and .exr).

Mitsuba's scene files cannot be loaded here; the wavefront harness renders its material ball instead
(wavefront.py).  Output: ``<out>.png`` (tone-mapped sRGB, written with the stdlib) and ``<out>.npy``
(linear radiance, the .exr's role).

    python -m bsdf_diffusion_sampling_amd.brdf_measured_disk --filename chm_orange_rgb --measured_dir tests/golden
    torchrun --nproc-per-node 8 -m bsdf_diffusion_sampling_amd.brdf_measured_spherical --filename aniso_miro_7_rgb
"""
from __future__ import annotations

import argparse
import os
import struct
import time
import zlib

import numpy as np
import torch


def write_png(path: str, rgb8: np.ndarray) -> None:
    """Minimal PNG writer (8-bit RGB, no dependencies)."""
    h, w, _ = rgb8.shape
    raw = b"".join(b"\x00" + rgb8[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def tonemap(img: np.ndarray) -> np.ndarray:
    """Reinhard + sRGB gamma -> uint8."""
    x = np.clip(img, 0, None)
    x = x / (1.0 + x)
    x = np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(np.maximum(x, 1e-12), 1 / 2.4) - 0.055)
    return (np.clip(x, 0, 1) * 255 + 0.5).astype(np.uint8)


def main(plugin_cls, default_out: str) -> None:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--filename", default="aniso_miro_7_rgb", help="material (the reference's props['filename'])")
    ap.add_argument("--scene_file", default=None,
                    help="a reference scene (rendering/matpreview/disney_bsdf_array*.xml, the reference CLI's "
                         "--scene_file): its `mybsdf` materials and ball positions are rendered as an array scene")
    ap.add_argument("--measured_dir", default=None, help="directory with <filename>.bsdf: render with the ground-truth f")
    ap.add_argument("--spp", type=int, default=4, help="samples per pixel per pass (the reference: SPP = 4)")
    ap.add_argument("--passes", type=int, default=128, help="number of passes (the reference: 128)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--albedo", type=float, nargs=3, default=[1.0, 1.0, 1.0])
    ap.add_argument("--out", default=default_out)
    a = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:  # image rows split over the ranks (one process per GPU), film tiles gathered on rank 0
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
        dist.init_process_group(os.environ.get("BSDFD_BENCH_BACKEND", "nccl"))
    from . import wavefront as WF
    if a.scene_file:
        from .materials import MaterialTable
        from .measured import MeasuredBSDF, find_measured_file
        names, cam, centers, radii = WF.scene_from_matpreview_xml(a.scene_file, a.size * 4 // 3, a.size)
        tab = MaterialTable([n + "_" + plugin_cls.DOMAIN_NAME for n in names])
        gts = {}
        if a.measured_dir:
            for i, n in enumerate(names):
                f = find_measured_file(n, a.measured_dir)
                if f:
                    gts[i] = MeasuredBSDF(f)
        r = WF.ArrayRenderer(tab, centers, radii, camera=cam, albedo=a.albedo, ground_truth=gts)
    else:
        props = {"filename": a.filename, "albedo": a.albedo}
        if a.measured_dir:
            props["measured_dir"] = a.measured_dir
        r = WF.WavefrontRenderer(plugin_cls(props), WF.Camera(width=a.size, height=a.size))
    t0 = time.time()
    img = r.render_sharded(a.passes, a.spp, seed=a.seed)
    torch.cuda.synchronize()
    if img is not None:
        img = img.cpu().numpy()
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        np.save(a.out + ".npy", img)
        write_png(a.out + ".png", tonemap(img))
        gt = (f"ground-truth f for {len(r.ground_truth)} of {len(r.table)} materials" if a.scene_file and r.use_ground_truth
              else "ground-truth f" if r.use_ground_truth else "proxy f = albedo * pdf")
        print(f"Render time: {time.time() - t0:.3f} seconds ({img.shape[1]}x{img.shape[0]}, {a.passes} x {a.spp} spp, "
              f"{gt}, {world} GPU(s)) -> {a.out}.png/.npy")
    if world > 1:
        dist.destroy_process_group()
