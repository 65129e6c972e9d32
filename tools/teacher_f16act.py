#!/usr/bin/env python3
"""A/B of the teacher kernel's sigmoid in fp16 (v_exp_f16 + v_rcp_f16 on a half-precision pre-activation) against the shipped fp32
form: a patched COPY of csrc/flow32.hip -> build_ab/lib_teacher_f16act.so, then error against the fp64
oracle and kernel time of both, alternating.      python tools/teacher_f16act.py build | run"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "bsdf_diffusion_sampling_amd", "csrc")
OUT = os.path.join(ROOT, "build_ab")


def build():
    s = open(os.path.join(CS, "flow32.hip")).read()
    old = "                    for (int v = 0; v < 16; ++v) hv[mt][v] = z[mt][v] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z[mt][v]));"
    assert s.count(old) == 1
    s = s.replace(old, "                    for (int v = 0; v < 16; ++v) {\n"
                       "                        const _Float16 zh = (_Float16)z[mt][v];\n"
                       "                        const _Float16 sg = __builtin_amdgcn_rcph((_Float16)1.0f + __builtin_elementwise_exp2(zh));\n"
                       "                        hv[mt][v] = (float)(zh * sg);\n"
                       "                    }")
    src = os.path.join(OUT, "flow32_teacher_f16act.hip")
    open(src, "w").write(s)
    common = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed", "-I", os.path.join(ROOT, "include"), "-I", CS]
    subprocess.run(["hipcc", *common, "-c", src, "-o", os.path.join(OUT, "flow32_teacher_f16act.o")], check=True)
    objs = [os.path.join(OUT, f"{t}.o") for t in ("bsdfd", "wavefront", "encoding", "measured", "bucket", "clock")]
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, os.path.join(OUT, "flow32_teacher_f16act.o"), "-o",
                    os.path.join(OUT, "lib_teacher_f16act.so")], check=True)
    os.remove(src)
    print("built")


def run():
    for rnd in range(2):
        for name, lib in (("fp32 sigmoid (shipped)", os.path.join(ROOT, "bsdf_diffusion_sampling_amd", "libbsdfd.so")),
                          ("fp16 sigmoid", os.path.join(OUT, "lib_teacher_f16act.so"))):
            env = dict(os.environ, BSDFD_LIB_PATH=lib, BSDFD_TILE="32")
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "teacher_check.py")], capture_output=True, text=True, env=env, timeout=600)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{") or l.startswith("32 ")]
            print(name, "|", " | ".join(lines) if lines else r.stderr[-300:], flush=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
