#!/usr/bin/env python3
"""Where the ENERGY of the disk Euler step goes (the flow kernel is power-limited): ablated COPIES of csrc/flow32.hip
(the product source is not touched) with one class of work removed — results are garbage, timing and clock are what is measured.

    python tools/energy_ablate.py build     # -> build_ab/lib_abl_<name>.so   (compiles build_ab/*.o through tools/ab_build32.sh if missing)
    python tools/energy_ablate.py run       # on the GPU box: kernel time, in-kernel clock, socket power (hwmon of this GPU) per variant
    python tools/energy_ablate.py           # both: the table of DESIGN.md §0 / profiles/r0N_ab/energy_ablate.jsonl in one command

Variants: full | nomfma (every v_mfma_f32_32x32x16_f16 replaced by an opaque pass-through of its accumulator) | notrans (v_exp_f32 /
v_rcp_f32 of the sigmoids replaced by one plain VALU op each) | nosplit (the hi/lo splits replaced by a plain fp16 pack: no v_and,
no lo part) | nomfma_notrans.  Under a power cap, time x power = energy; a variant that is no longer capped shows it by its clock."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "bsdf_diffusion_sampling_amd", "csrc")
OUT = os.path.join(ROOT, "build_ab")
VARIANTS = ["full", "nomfma", "notrans", "nosplit", "nomfma_notrans"]


def patched(name):
    s = open(os.path.join(CS, "flow32.hip")).read()

    def rep(old, new):
        nonlocal s
        assert s.count(old) == 1, (name, old[:60], s.count(old))
        s = s.replace(old, new)
    if "nomfma" in name:
        rep("    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);",
            '    asm volatile("" : "+v"(c) : "v"(a), "v"(b));\n    return c;')
    if "notrans" in name:
        # flow_dev.h's silu_grad_scaled is shared with bsdfd.hip: shadow it in this translation unit
        rep("template <bool WITH_G>\n__device__ __forceinline__ void act16(",
            "__device__ __forceinline__ void silu_grad_scaled_abl(float zs, float& hs, float& g) {\n"
            "    const float s = 0.3f * (1.0f + 0.5f * zs);\n    hs = zs * s;\n    g = fmaf(hs, fmaf(s, kLn2, -kLn2), s);\n}\n"
            "template <bool WITH_G>\n__device__ __forceinline__ void act16(")
        rep("        if (WITH_G) silu_grad_scaled(z[v], hs[v], g[v]);", "        if (WITH_G) silu_grad_scaled_abl(z[v], hs[v], g[v]);")
    if "nosplit" in name:
        rep("            split_pack<SPLIT>(x4, hi[c].p[2 * k], hi[c].p[2 * k + 1], lo[c].p[2 * k], lo[c].p[2 * k + 1]);",
            "            split_pack<false>(x4, hi[c].p[2 * k], hi[c].p[2 * k + 1], lo[c].p[2 * k], lo[c].p[2 * k + 1]);\n"
            "            lo[c].p[2 * k] = hi[c].p[2 * k]; lo[c].p[2 * k + 1] = hi[c].p[2 * k + 1];")
    return s


def build():
    os.makedirs(OUT, exist_ok=True)
    if not all(os.path.exists(os.path.join(OUT, f"{t}.o")) for t in ("bsdfd", "wavefront", "encoding", "measured", "bucket", "clock")):
        subprocess.run(["bash", os.path.join(ROOT, "tools", "ab_build32.sh"), "base", ""], check=True)
    common = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed", "-I", os.path.join(ROOT, "include"), "-I", CS]
    procs = []
    for v in VARIANTS:
        src = os.path.join(OUT, f"flow32_abl_{v}.hip")
        open(src, "w").write(patched(v))
        procs.append((v, subprocess.Popen(["hipcc", *common, "-c", src, "-o", os.path.join(OUT, f"flow32_abl_{v}.o")])))
    for v, pr in procs:
        assert pr.wait() == 0, v
        objs = [os.path.join(OUT, f"{t}.o") for t in ("bsdfd", "wavefront", "encoding", "measured", "bucket", "clock")]
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, os.path.join(OUT, f"flow32_abl_{v}.o"), "-o",
                        os.path.join(OUT, f"lib_abl_{v}.so")], check=True)
        os.remove(os.path.join(OUT, f"flow32_abl_{v}.hip"))
        print("built", v)


CHILD = r'''
import sys, time, subprocess, threading, json
sys.path.insert(0, %(root)r)
import torch
import bench
from bsdf_diffusion_sampling_amd import weights as W
from bsdf_diffusion_sampling_amd.sampler import FlowSampler
n, T = 1 << 20, 8
dev = torch.device("cuda")
s = FlowSampler(W.load(W.shipped_path("aniso_miro_7_rgb", "disk")), tile=32)
wi = bench.make_wi("disk", n, 1234, dev)
wo = torch.empty((n, 3), device=dev); p = torch.empty(n, device=dev)
from bsdf_diffusion_sampling_amd.power import PowerSampler   # hwmon of THIS GPU (a node's sysfs shows all eight), rocm-smi as fallback
t0 = time.time()
while time.time() - t0 < 1.0:
    s.plugin_sample(wi, None, T=T, out=(wo, p)); torch.cuda.synchronize()
ps = PowerSampler(); ps.start()
s.set_profiling(True)
t0 = time.time()
while time.time() - t0 < 3.0:
    for _ in range(20): s.plugin_sample(wi, None, T=T, out=(wo, p))
    torch.cuda.synchronize()
k, ms = s.profile_read(); mhz = s.profile_clock_mhz(); s.set_profiling(False)
pw = ps.stop()
w = pw["socket_power_w"]
print(json.dumps({"variant": %(name)r, "us_per_launch": ms / k * 1e3, "in_kernel_mhz": mhz, "kcycles": ms / k * mhz,
                  "socket_w_median": w, "sclk": pw["sclk_mhz"], "power_source": pw["source"], "power_samples": pw["samples"],
                  "joule_per_launch": (w * ms / k * 1e-3) if w else None}))
'''


def run():
    for rnd in range(2):
        for v in VARIANTS:
            env = dict(os.environ, BSDFD_LIB_PATH=os.path.join(OUT, f"lib_abl_{v}.so"))
            r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "name": v}], capture_output=True, text=True, env=env, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(line[-1] if line else ("FAILED " + v + " " + r.stderr[-300:]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) < 2:
        build()
        run()
    else:
        {"build": build, "run": run}[sys.argv[1]]()
