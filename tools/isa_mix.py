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
import collections
import json
import re
import sys

# Occupancy of a SIMD's VALU issue path, shader cycles per wave64 instruction (tools/ubench/RESULTS.md, "Round 3").
# Round 3 replaced the additive "MFMA time + VALU time" model: PMC passes of tools/ubench/spec2 show that a
# v_mfma_f32_16x16x32_f16 holds the VALU issue path for ~9.6 of its 16 matrix-pipe cycles (SQ_ACTIVE_INST_VALU counts 2.4
# quad-cycles per MFMA), and tools/ubench/bank shows what the VALU classes cost at 3-4 waves/SIMD: plain VOP3 2.56, VOP2
# 2.14, packed-f32 / converting 4.29, a transcendental 8.1 in a pure stream but ~11 between plain instructions (the
# sigmoid chains exp -> add -> rcp -> mul measure 10.5 batched, 11.4 unit by unit).
COST = {"v_mfma_f32_16x16x32_f16": 9.6, "v_mfma_f32_16x16x4_f32": 19.2, "v_mfma_f32_32x32x16_f16": 19.2,
        "v_exp_f32": 11.0, "v_rcp_f32": 11.0, "v_log_f32": 11.0, "v_sqrt_f32": 11.0, "v_sin_f32": 11.0, "v_cos_f32": 11.0,
        "v_rsq_f32": 11.0, "v_exp_f16": 11.0, "v_rcp_f16": 11.0,
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


def kernel_body(lines, key):
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and key in l and l.split(";")[0].rstrip().endswith(":"):
            start = i
            break
    if start is None:
        raise SystemExit(f"kernel matching {key!r} not found")
    for j in range(start, len(lines)):
        if lines[j].strip().startswith("s_endpgm"):
            return lines[start:j + 1]
    return lines[start:]


def loops(body):
    """(first, last) line indices of every label .. backward-branch pair."""
    labels = {}
    out = []
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
        m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels:
            out.append((labels[m.group(1)], i))
    return out


def mix(body, lo, hi):
    c = collections.Counter()
    for l in body[lo:hi + 1]:
        l = l.split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."):
            continue
        c[re.sub(r"_(e32|e64|sdwa|dpp)$", "", l.split()[0])] += 1
    return c


KERNEL_OF_WORKLOAD = {  # bench.py workload -> mangled-name fragment of its flow kernel (split3, Jacobian)
    "disk_1Mi_T8": "flow_kernelILi0ELi2ELi2ELb1ELi3ELb0E",
    "disk_1Mi_T4": "flow_kernelILi0ELi2ELi2ELb1ELi3ELb0E",
    "spherical_16Mi_T8": "flow_kernelILi1ELi2ELi2ELb1ELi4ELb0E",
}


def profile(out_path):
    """Compile the shipped kernel source to assembly and write the per-workload models bench.py reads
    (profiles/isa_mix_latest.json)."""
    import os
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "bsdfd.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-value",
                        "-Wno-pass-failed", "-Wno-unused-command-line-argument", "-I", os.path.join(root, "include"),
                        os.path.join(root, "bsdf_diffusion_sampling_amd", "csrc", "bsdfd.hip"), "-o", asm], check=True)
        res = {w: model(asm, k) for w, k in KERNEL_OF_WORKLOAD.items()}
    import hashlib
    src = os.path.join(root, "bsdf_diffusion_sampling_amd", "csrc", "bsdfd.hip")
    git = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "-C", root, "status", "--porcelain", "--", src], capture_output=True, text=True).stdout.strip())
    res["_meta"] = {"kernel_source_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest(), "git": git + ("+dirty" if dirty else ""),
                    "tool": "tools/isa_mix.py --profile", "cost_model": "VALU-issue-path occupancy, round 3 (see COST in tools/isa_mix.py)"}
    json.dump(res, open(out_path, "w"), indent=1)
    print(f"wrote {out_path}")


def _regs(tok):
    """VGPR indices named by an operand token: v12 -> {12}, v[4:7] -> {4..7}."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def check_async(path, key):
    """The flow kernel fetches weight fragments with inline-asm `ds_read_b128` whose destinations the compiler believes valid
    at once (csrc/bsdfd.hip, lds_read_b128_async_at); they are only valid after the next `s_waitcnt lgkmcnt(0)`.  Verify on the
    built assembly that no instruction reads or writes a destination register in between, and that the kernel has no scratch
    (a spill of such a register would store stale data).  Returns the list of violations (empty = safe)."""
    lines = open(path).read().splitlines()
    body = kernel_body(lines, key)
    bad, pending, in_asm, n_async = [], {}, False, 0
    for i, raw in enumerate(body):
        l = raw.strip()
        if l.startswith(";;#ASMSTART") or l.startswith(";#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND") or l.startswith(";#ASMEND"):
            in_asm = False
            continue
        code = l.split(";")[0].strip()
        if not code or code.endswith(":") or code.startswith("."):
            continue
        op = code.split()[0]
        toks = [t.strip(",") for t in code.split()[1:]]
        if op.startswith("s_waitcnt") and "lgkmcnt(0)" in code:
            pending.clear()
            continue
        touched = set().union(*[_regs(t) for t in toks]) if toks else set()
        for r in touched & set(pending):
            bad.append(f"line {i}: `{code}` touches v{r}, the destination of the asynchronous read at line {pending[r]}")
        if in_asm and op == "ds_read_b128":
            n_async += 1
            for r in _regs(toks[0]):
                pending[r] = i
        if op.startswith("s_cbranch") or op.startswith("s_branch") or op == "s_endpgm":
            if pending:
                bad.append(f"line {i}: control flow `{code}` with {len(pending)} asynchronous destination registers still pending")
                pending.clear()
    meta = {}
    for i, l in enumerate(lines):
        if ".name:" in l and key in l:
            for l2 in lines[i:i + 16]:
                m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", l2)
                if m:
                    meta["scratch"] = int(m.group(1))
                m = re.search(r"\.vgpr_count:\s+(\d+)", l2)
                if m:
                    meta["vgprs"] = int(m.group(1))
            break
    if meta.get("scratch", 0) != 0:
        bad.append(f"kernel uses {meta['scratch']} B/lane of scratch: a spilled asynchronous destination would be stale")
    if meta.get("vgprs", 0) > 168:
        bad.append(f"kernel uses {meta['vgprs']} VGPRs: more than the 168 that 3 waves/SIMD allow (csrc/bsdfd.hip, BSDFD_MIN_WAVES)")
    return n_async, bad


def main():
    if sys.argv[1] == "--profile":
        return profile(sys.argv[2])
    if sys.argv[1] == "--check-async":
        import os
        import subprocess
        import tempfile
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        with tempfile.TemporaryDirectory() as td:
            asm = os.path.join(td, "bsdfd.s")
            subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-value",
                            "-Wno-pass-failed", "-Wno-unused-command-line-argument", "-I", os.path.join(root, "include"),
                            os.path.join(root, "bsdf_diffusion_sampling_amd", "csrc", "bsdfd.hip"), "-o", asm], check=True)
            rc = 0
            for key in ("flow_kernelILi0ELi2ELi2ELb1ELi3ELb0E", "flow_kernelILi0ELi2ELi3ELb1ELi3ELb0E",    # disk 32x3: split3, f16
                        "flow_kernelILi0ELi2ELi2ELb1ELi3ELb1E", "flow_kernelILi0ELi2ELi3ELb1ELi3ELb1E",    # ... fused sample+pdf
                        "flow_kernelILi1ELi2ELi2ELb1ELi4ELb0E", "flow_kernelILi1ELi2ELi3ELb1ELi4ELb0E"):   # spherical 32x4
                n, bad = check_async(asm, key)
                print(f"{key}: {n} asynchronous ds_read_b128, {len(bad)} violations")
                for b in bad:
                    print("  ", b)
                rc |= bool(bad) or n == 0
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
        "kernel": key, "loop_lines": best[1] - best[0] + 1,
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
    # the step is bound by the VALU issue path (MFMA issue occupancy + VALU), never by the matrix pipe itself here
    res["issue_cycles_total"] = round(max(res["issue_cycles_mfma"] + res["issue_cycles_valu"], res["matrix_pipe_cycles"]), 1)
    return res


if __name__ == "__main__":
    sys.exit(main() or 0)
