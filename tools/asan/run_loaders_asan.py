#!/usr/bin/env python3
"""CPU-only: the two file parsers of the C ABI (bsdfd_measured_create_from_file: RGL tensor files;
bsdfd_create_from_file: .bsdfw weights) compiled for the HOST with the address sanitizer and fed malformed files.
Every case must be refused with BSDFD_EIO and the sanitizer must stay silent.  Run in the build container:

    python tools/asan/run_loaders_asan.py

(GPU sanitizer builds are not available on the GPU pool, and gpurun refuses snapshots whose tests would build one, so
this lives under tools/ — listed in .gpurunignore — and not under tests/.)  Last run: round 2, 11 cases, clean.
"""
import os
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bsdf_diffusion_sampling_amd import weights as W  # noqa: E402

SAN = "-fsanitize=address"  # host code only: -fno-gpu-sanitize below keeps the device code uninstrumented


def main():
    tmp = tempfile.mkdtemp()
    csrc = os.path.join(ROOT, "bsdf_diffusion_sampling_amd", "csrc")
    exe = os.path.join(tmp, "loaders_asan")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", SAN, "-fno-gpu-sanitize", "-fno-omit-frame-pointer",
                    "-Wno-unused-value", "-Wno-pass-failed", "-I", os.path.join(ROOT, "include"), os.path.join(csrc, "measured.hip"),
                    os.path.join(csrc, "bsdfd.hip"), os.path.join(csrc, "flow32.hip"), os.path.join(ROOT, "tools", "asan", "loaders_asan.cpp"), "-o", exe], check=True)
    raw = open(os.path.join(ROOT, "tests", "golden", "chm_orange_rgb.bsdf"), "rb").read()
    nf = struct.unpack_from("<I", raw, 14)[0]
    pos, rec = 18, {}
    for _ in range(nf):
        nl = struct.unpack_from("<H", raw, pos)[0]
        name = raw[pos + 2: pos + 2 + nl].decode()
        nd = struct.unpack_from("<H", raw, pos + 2 + nl)[0]
        rec[name] = (nd, pos + 2 + nl + 2 + 1 + 8)
        pos = rec[name][1] + 8 * nd

    def patched(name, dims, offset=None):
        b = bytearray(raw)
        nd, at = rec[name]
        struct.pack_into("<%dQ" % nd, b, at, *dims)
        if offset is not None:
            struct.pack_into("<Q", b, at - 8, offset)
        return bytes(b)
    w = open(W.shipped_path("chm_orange_rgb", "disk"), "rb").read()
    wb = bytearray(w)
    struct.pack_into("<i", wb, 76, 1 << 30)  # absurd width in the header
    cases = {"zero.bsdf": patched("theta_i", [0]), "wrap.bsdf": patched("vndf", [1 << 26] * 4),
             "big.bsdf": patched("ndf", [1 << 20, 1 << 20]), "off.bsdf": patched("rgb", [1, 8, 3, 32, 32], len(raw) - 16),
             "trunc.bsdf": raw[: len(raw) // 2], "hdr.bsdf": raw[:40], "junk.bsdf": b"not a tensor file at all",
             "trunc.bsdfw": w[:200], "width.bsdfw": bytes(wb), "short.bsdfw": w[:50], "tail.bsdfw": w + b"xx"}
    for k, v in cases.items():
        open(os.path.join(tmp, k), "wb").write(v)
    r = subprocess.run([exe] + [os.path.join(tmp, k) for k in cases], capture_output=True, text=True,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"), timeout=120)
    print(r.stdout)
    assert r.returncode == 0 and "AddressSanitizer" not in r.stderr, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if "-> rc=" in l]
    assert len(lines) == 2 * len(cases) and all("rc=3" in l for l in lines), r.stdout
    print(f"{len(cases)} malformed files refused, sanitizer silent")


if __name__ == "__main__":
    main()
