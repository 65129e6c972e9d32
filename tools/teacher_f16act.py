#!/usr/bin/env python3
"""A/B of the teacher kernel's fp16 sigmoids as the COMPILER writes them (value by value: v_cvt_f16_f32, v_exp_f16, v_add_f16,
v_rcp_f16, v_mul_f16, v_pack_b32_f16) against the shipped hand-packed form (csrc/flow32.hip: act_pack8 / BSDFD_PK4_TRANS): a patched
COPY of csrc/flow32.hip -> build_ab/lib_teacher_f16act.so, then error against the fp64 oracle and kernel time of both, alternating.
History: against the fp32 sigmoids this compiler form was +4 % (profiles/r05_ab/teacher_f16_sigmoid.txt);
the hand-packed form is -8.6 % (profiles/r05_ab/teacher_packed_f16_sigmoid.txt).      python tools/teacher_f16act.py build | run"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "bsdf_diffusion_sampling_amd", "csrc")
OUT = os.path.join(ROOT, "build_ab")


def build():
    s = open(os.path.join(CS, "flow32.hip")).read()
    old = "                    act_pack8(z8, fr[c]);"
    assert s.count(old) == 1, "the teacher kernel of csrc/flow32.hip has changed: update `old`"
    # what hipcc makes of the same arithmetic written value by value: v_cvt_f16_f32, v_exp_f16, v_add_f16, v_rcp_f16, v_mul_f16,
    # v_pack_b32_f16 - no packed math, no SDWA halves
    s = s.replace(old, "                    for (int k = 0; k < 4; ++k) {\n"
                       "                        const _Float16 z0 = (_Float16)z8[2 * k], z1 = (_Float16)z8[2 * k + 1];\n"
                       "                        const _Float16 s0 = __builtin_amdgcn_rcph((_Float16)1.0f + __builtin_elementwise_exp2(z0));\n"
                       "                        const _Float16 s1 = __builtin_amdgcn_rcph((_Float16)1.0f + __builtin_elementwise_exp2(z1));\n"
                       "                        fr[c].p[k] = (f16x2){z0 * s0, z1 * s1};\n"
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
        for name, lib in (("packed fp16 sigmoid (shipped)", os.path.join(ROOT, "bsdf_diffusion_sampling_amd", "libbsdfd.so")),
                          ("fp16 sigmoid, compiler form", os.path.join(OUT, "lib_teacher_f16act.so"))):
            env = dict(os.environ, BSDFD_LIB_PATH=lib, BSDFD_TILE="32")
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "teacher_check.py")], capture_output=True, text=True, env=env, timeout=600)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{") or l.startswith("32 ")]
            print(name, "|", " | ".join(lines) if lines else r.stderr[-300:], flush=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
