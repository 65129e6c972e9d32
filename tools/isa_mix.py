#!/usr/bin/env python3
"""Instruction mix of the Euler-step loop of one flow_kernel instantiation, from the gfx950 assembly.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -I include \
        bsdf_diffusion_sampling_amd/csrc/bsdfd.hip -o /tmp/bsdfd.s
    python tools/isa_mix.py /tmp/bsdfd.s 'flow_kernelILi0ELi2ELi2ELb1ELi3ELb0E'

Finds the kernel's body, takes the innermost loop that holds the most MFMAs (the Euler step: a label
and the backward branch to it) and prints the count per mnemonic, the issue-cycle estimate at the rates
of tools/ubench/RESULTS.md, and the kernel's register / scratch footprint.  Used for the
"VALU instructions per tile-step" column of DESIGN.md §4 and the issue-bound roofline of bench.py.
"""
import json
import re
import sys

# What an instruction costs the SIMD, shader cycles per wave64 instruction at 3-4 waves/SIMD (tools/ubench/RESULTS.md).
# Round 4 (tools/ubench/mfma_src, profiles/r04_ab/): in a VALU-heavy stream — this kernel issues ~7 VALU per MFMA — MFMA time
# and VALU time ADD: [1 MFMA + 8 v_fma] costs 36.4 cycles = 16.4 + 8 x 2.5 whatever the operands' register file (VGPR / AGPR /
# inline 0), the order of dependent MFMAs, or the number of co-resident waves; only up to ~2 plain VALU per MFMA hide in an
# MFMA-bound stream.  (Round 3 had priced an MFMA at the 9.6 cycles SQ_ACTIVE_INST_VALU attributes to it; that counter counts
# issue events in quad-cycle granules, not occupancy: it reads 97 % on a stream that is saturated by construction and would
# read 167 % on pure v_fma.)  VALU classes: plain VOP3 2.56, VOP2 2.14, packed-f32 / converting 4.29, a transcendental 8.1 (its
# rate in a pure stream, tools/ubench/valu_rates2).  Rounds 2-4 priced it at the ~11 a micro-benchmark read between plain
# instructions (tools/ubench/bank); round 5 measured it IN the kernels three ways — all 97 of the disk step replaced by one
# plain instruction each: -14 % cycles = 7.4 apiece (profiles/r05_ab/energy_ablate.jsonl); 24 v_rcp_f32 traded for 47 plain
# instructions: +3 % time where 11 predicted -3 % (ab32_rcp_pairs.txt); the teacher's step, 384 of them, 5 240 cycles measured
# against 6 715 by addition at 11 — and with 8.1 every kernel's measured step is within 8 % of its sum (profiles/HISTORY.md, appendix §4.6).
COST = {"v_mfma_f32_16x16x32_f16": 16.35, "v_mfma_f32_16x16x4_f32": 32.0, "v_mfma_f32_32x32x16_f16": 32.1,
        "v_exp_f32": 8.1, "v_rcp_f32": 8.1, "v_log_f32": 8.1, "v_sqrt_f32": 8.1, "v_sin_f32": 8.1, "v_cos_f32": 8.1,
        "v_rsq_f32": 8.1, "v_exp_f16": 8.1, "v_rcp_f16": 8.1,
        "v_cvt_pk_f16_f32": 4.29, "v_cvt_pkrtz_f16_f32": 4.4, "v_perm_b32": 4.3, "v_cvt_f32_f16": 4.2,
        "v_fma_mix_f32": 7.3, "v_fma_mixlo_f16": 7.3, "v_fma_mixhi_f16": 7.3,
        "v_mul_f32": 2.14, "v_add_f32": 2.14, "v_sub_f32": 2.14, "v_permlane32_swap_b32": 8.2, "v_permlane16_swap_b32": 8.2}
PK = 4.29     # v_pk_{fma,mul,add}_f32
PLAIN = 2.56  # any other VALU
MATRIX_PIPE = {"v_mfma_f32_16x16x32_f16": 16.2, "v_mfma_f32_16x16x4_f32": 32.0, "v_mfma_f32_32x32x16_f16": 32.1}  # matrix-pipe cycles


def cost(m):
    if m in COST:
        return COST[m]
    if m.startswith("v_pk_") and m.endswith("_f32"):
        return PK
    if m.startswith("v_"):
        return PLAIN
    return 0.0


import os as _os
sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from bsdf_diffusion_sampling_amd._asmcheck import check_async_lines, check_file_mfma, kernel_body, loops, mix  # noqa: E402


KERNEL_OF_WORKLOAD = {  # bench.py workload -> mangled-name fragment of its flow kernel (split3, Jacobian; the library's default tile)
    "disk_1Mi_T8": "flow_kernel32ILi0ELb1ELb0E",
    "disk_1Mi_T4": "flow_kernel32ILi0ELb1ELb0E",
    "spherical_16Mi_T8": "flow_kernel32ILi1ELb1ELb0E",
    "teacher_64x6_4Mi_T128": "flow_kernel32wE",
    "teacher_64x6_4Mi_T128@tile16": "flow_kernelILi1ELi4ELi3ELb0ELi6ELb0E",
    "complex64_1Mi_T8": "flow_kernelILi1ELi4ELi2ELb1ELi6ELb0E",
    # the 16-query-tile kernels of the same nets (bsdfd_desc.tile = 16), for comparison
    "disk_1Mi_T8@tile16": "flow_kernelILi0ELi2ELi2ELb1ELi3ELb0E",
    "disk_1Mi_T4@tile16": "flow_kernelILi0ELi2ELi2ELb1ELi3ELb0E",
    "spherical_16Mi_T8@tile16": "flow_kernelILi1ELi2ELi2ELb1ELi4ELb0E",
}


def tile_of(key):
    """Queries per wave tile of a kernel (name fragment): 32 for csrc/flow32.hip, 16 for csrc/bsdfd.hip."""
    return 32 if "flow_kernel32" in key else 16


def compile_asm(td):
    """Device assembly of the two flow-kernel translation units -> {"bsdfd": path, "flow32": path} (compiled in parallel)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs, out = [], {}
    for stem in ("bsdfd", "flow32"):
        out[stem] = os.path.join(td, stem + ".s")
        procs.append(subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-value",
                                       "-Wno-pass-failed", "-Wno-unused-command-line-argument", "-I", os.path.join(root, "include"),
                                       os.path.join(root, "bsdf_diffusion_sampling_amd", "csrc", stem + ".hip"), "-o", out[stem]]))
    for pr in procs:
        if pr.wait() != 0:
            raise RuntimeError("hipcc failed")
    return out


def asm_of(paths, key):
    return paths["flow32" if "flow_kernel32" in key else "bsdfd"]


def profile(out_path):
    """Compile the shipped kernel source to assembly and write the per-workload models bench.py reads
    (profiles/isa_mix_latest.json)."""
    import os
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        paths = compile_asm(td)
        res = {w: model(asm_of(paths, k), k) for w, k in KERNEL_OF_WORKLOAD.items()}
    sys.path.insert(0, root)
    from bsdf_diffusion_sampling_amd import _lib
    git = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "-C", root, "status", "--porcelain", "--", *_lib.KERNEL_SOURCES], capture_output=True, text=True).stdout.strip())
    res["_meta"] = {"kernel_source_sha256": _lib.kernel_source_sha256(), "git": git + ("+dirty" if dirty else ""),
                    "tool": "tools/isa_mix.py --profile", "cost_model": "additive: sum of VALU costs + matrix-pipe time of the MFMAs, round 4 (see COST in tools/isa_mix.py)"}
    json.dump(res, open(out_path, "w"), indent=1)
    print(f"wrote {out_path}")


def check_async(path, key):
    """(asynchronous reads, violations) of one kernel of a built assembly file: bsdf_diffusion_sampling_amd/_asmcheck.py (the
    check `_lib.build()` runs on its own artefact), plus the occupancy guard of the headline kernel: more than 168 VGPRs would
    drop the 32-wide kernels from 3 to 2 waves/SIMD (csrc/bsdfd.hip, BSDFD_MIN_WAVES; -4 % when measured)."""
    lines = open(path).read().splitlines()
    n, bad = check_async_lines(lines, key)
    for i, l in enumerate(lines):
        if ".name:" in l and key in l:
            for l2 in lines[i:i + 16]:
                m = re.search(r"\.vgpr_count:\s+(\d+)", l2)
                if m and int(m.group(1)) > 168 and "ILi2ELi" in key:
                    bad.append(f"kernel uses {m.group(1)} VGPRs: more than the 168 that 3 waves/SIMD allow (csrc/bsdfd.hip, BSDFD_MIN_WAVES)")
            break
    return n, bad


def main():
    if sys.argv[1] == "--profile":
        return profile(sys.argv[2])
    if sys.argv[1] == "--check-async":
        import os
        import subprocess
        import tempfile
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        with tempfile.TemporaryDirectory() as td:
            paths = compile_asm(td)
            asm = paths["bsdfd"]
            rc = 0
            for key in ("flow_kernelILi0ELi2ELi2ELb1ELi3ELb0E", "flow_kernelILi0ELi2ELi3ELb1ELi3ELb0E",    # disk 32x3: split3, f16
                        "flow_kernelILi0ELi2ELi2ELb1ELi3ELb1E", "flow_kernelILi0ELi2ELi3ELb1ELi3ELb1E",    # ... fused sample+pdf
                        "flow_kernelILi1ELi2ELi2ELb1ELi4ELb0E", "flow_kernelILi1ELi2ELi3ELb1ELi4ELb0E",    # spherical 32x4
                        "flow_kernelILi1ELi2ELi2ELb1ELi4ELb1E", "flow_kernelILi1ELi2ELi3ELb1ELi4ELb1E"):   # ... fused sample+pdf
                n, bad = check_async(asm, key)
                print(f"{key}: {n} asynchronous ds_read_b128, {len(bad)} violations")
                for b in bad:
                    print("  ", b)
                rc |= bool(bad) or n == 0
            from bsdf_diffusion_sampling_amd._asmcheck import verify_census
            problems = verify_census([paths["bsdfd"], paths["flow32"]], "async")
            print(f"census of the two translation units: {len(problems)} problem(s)")
            for m in problems[:8]:
                print("  ", m)
            rc |= bool(problems)
            hz = check_file_mfma(asm)
            hz.update(check_file_mfma(paths["flow32"]))
            n_bad = sum(1 for _, b in hz.values() if b)
            print(f"MFMA results consumed before their wait states: {n_bad} of {len(hz)} kernels, {sum(n for n, _ in hz.values())} MFMAs checked")
            for k, (_, b) in hz.items():
                for x in b[:2]:
                    print("  ", k, x)
            rc |= bool(n_bad)
            # occupancy of the kernels behind the tracked workloads (waves/SIMD the VGPR allocation allows): a prologue edit shared
            # by all instantiations once took the teacher sampler from 4 to 3 waves unnoticed (round 4, -4 %)
            from bsdf_diffusion_sampling_amd._asmcheck import kernel_meta
            lines = open(asm).read().splitlines() + open(paths["flow32"]).read().splitlines()
            for name, key, want in (("disk 32x3 split3, 32-query tiles (disk_1Mi_T8 / T4)", "flow_kernel32ILi0ELb1ELb0E", 3),
                                    ("spherical 32x4 split3, 32-query tiles (spherical_16Mi_T8)", "flow_kernel32ILi1ELb1ELb0E", 3),
                                    ("disk 32x3 fused sample+pdf, 32-query tiles", "flow_kernel32ILi0ELb1ELb1E", 3),
                                    ("spherical 32x4 fused sample+pdf, 32-query tiles", "flow_kernel32ILi1ELb1ELb1E", 2),
                                    ("disk 32x3 split3, 16-query tiles", "flow_kernelILi0ELi2ELi2ELb1ELi3ELb0E", 3),
                                    ("spherical 32x4 split3, 16-query tiles", "flow_kernelILi1ELi2ELi2ELb1ELi4ELb0E", 3),
                                    ("teacher 64x6 f16, no Jacobian, 32-query tiles", "flow_kernel32wE", 4),
                                    ("teacher 64x6 f16, no Jacobian, 16-query tiles", "flow_kernelILi1ELi4ELi3ELb0ELi6ELb0E", 4),
                                    ("64x6 split3 with Jacobian (complex64_1Mi_T8)", "flow_kernelILi1ELi4ELi2ELb1ELi6ELb0E", 2)):
                v = kernel_meta(lines, key).get("vgpr_count", 0)
                waves = 8 if v <= 64 else 512 // ((v + 7) // 8 * 8)
                print(f"occupancy {name}: {v} VGPRs = {waves} waves/SIMD (wanted {want}){'' if waves >= want else '  <-- LOST'}")
                rc |= waves < want
            from bsdf_diffusion_sampling_amd._asmcheck import check_file_swap
            sw = check_file_swap(asm)
            sw.update(check_file_swap(paths["flow32"]))
            n_bad = sum(1 for _, b in sw.values() if b)
            print(f"lane swaps of a register written fewer than 2 wait states earlier: {n_bad} of {len(sw)} kernels, {sum(n for n, _ in sw.values())} swaps checked")
            rc |= bool(n_bad)
        return rc
    print(json.dumps(model(sys.argv[1], sys.argv[2]), indent=1))


def model(path, key):
    lines = open(path).read().splitlines()
    body = kernel_body(lines, key)
    # the Euler-step loop = the SHORTEST loop that holds at least half of the MFMAs of the MFMA-richest loop
    # (the tile loop around it holds the prologue's MFMAs as well)
    cand = []
    for lo, hi in loops(body):
        c = mix(body, lo, hi)
        cand.append((sum(v for k, v in c.items() if k.startswith("v_mfma")), hi - lo, (lo, hi)))
    top = max(n for n, _, _ in cand)
    best = min((span, rng) for n, span, rng in cand if 2 * n >= top)[1]
    c = mix(body, *best)
    mfma = {k: v for k, v in c.items() if k.startswith("v_mfma")}
    valu = {k: v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma")}
    other = {k: v for k, v in c.items() if not k.startswith("v_")}
    meta = {}
    # metadata block of this kernel
    for i, l in enumerate(lines):
        if ".name:" in l and key in l:
            for l2 in lines[i:i + 16]:
                for f in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "private_segment_fixed_size"):
                    m = re.search(rf"\.{f}:\s+(\d+)", l2)
                    if m:
                        meta[f] = int(m.group(1))
            break
    res = {
        "kernel": key, "tile_queries": tile_of(key), "loop_lines": best[1] - best[0] + 1,
        "mfma": mfma, "n_mfma": sum(mfma.values()),
        "n_valu": sum(valu.values()),
        "valu_top": dict(sorted(valu.items(), key=lambda kv: -kv[1])[:14]),
        "n_pk_f32": sum(v for k, v in valu.items() if k.startswith("v_pk_") and k.endswith("_f32")),
        "n_trans": sum(v for k, v in valu.items() if k in ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_sqrt_f32", "v_sin_f32",
                                                          "v_cos_f32", "v_rsq_f32", "v_exp_f16", "v_rcp_f16")),
        "other": other,
        "issue_cycles_mfma": round(sum(cost(k) * v for k, v in mfma.items()), 1),
        "issue_cycles_valu": round(sum(cost(k) * v for k, v in valu.items()), 1),
        "matrix_pipe_cycles": round(sum(MATRIX_PIPE.get(k, 16.2) * v for k, v in mfma.items()), 1),
        "meta": meta,
    }
    # MFMA time and VALU time add (round 4, tools/ubench/mfma_src)
    res["issue_cycles_total"] = round(res["issue_cycles_mfma"] + res["issue_cycles_valu"], 1)
    return res


if __name__ == "__main__":
    sys.exit(main() or 0)
