#!/usr/bin/env python3
"""CPU-only: the weight-image builders of the 32-query-tile kernels (csrc/flow32.hip) compiled for the HOST with the address
sanitizer and run on shipped weight sets of the nets they serve (disk 32x3 split3 / f16, spherical 32x4 split3, 64x6 f16 teacher, 64x6 split3 with the Jacobian).
The sanitizer must stay silent.      python tools/asan/run_image_asan.py
(tools/asan/ is listed in .gpurunignore: gpurun refuses snapshots whose tests would build a sanitizer binary.)"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bsdf_diffusion_sampling_amd import weights as W  # noqa: E402


def main():
    tmp = tempfile.mkdtemp()
    csrc = os.path.join(ROOT, "bsdf_diffusion_sampling_amd", "csrc")
    exe = os.path.join(tmp, "image_asan")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fsanitize=address", "-fno-gpu-sanitize", "-fno-omit-frame-pointer",
                    "-Wno-unused-value", "-Wno-pass-failed", "-I", os.path.join(ROOT, "include"), "-I", csrc, os.path.join(csrc, "flow32.hip"),
                    os.path.join(ROOT, "tools", "asan", "image_asan.cpp"), "-o", exe], check=True)
    args = []
    for material, domain, kind, prec in (("aniso_miro_7_rgb", "disk", None, 2), ("chm_orange_rgb", "disk", None, 2),
                                        ("aniso_miro_7_rgb", "spherical", None, 2), ("bsdf_3", "spherical", None, 2),
                                        ("aniso_miro_7_rgb", "spherical", "complex", 3), ("aniso_miro_7_rgb", "spherical", "complex", 2),
                                        ("aniso_miro_7_rgb", "disk", "diffusion", 3)):
        args += [W.shipped_path(material, domain, kind) if kind else W.shipped_path(material, domain), str(prec)]
    r = subprocess.run([exe] + args, capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"), timeout=300)
    print(r.stdout)
    assert r.returncode == 0 and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.stdout.count("image ") == 7 and r.stdout.count("no 32-query-tile kernel") == 0, r.stdout   # (round 6: the 64 x 6 split3 net has flow_kernel32c)
    print("clean")


if __name__ == "__main__":
    main()
