"""Safety check of the flow kernels' asynchronous LDS reads on the gfx950 assembly of a build (pure text analysis, no GPU).

csrc/bsdfd.hip fetches weight fragments with inline-asm ``ds_read_b128`` whose destination registers the compiler believes
valid at once; they only become valid at the following ``s_waitcnt lgkmcnt(0)``.  Whether the code between a read and its
wait leaves those registers alone is a property of the compiler's register allocation and scheduling, i.e. of the toolchain
that builds the library — so ``_lib.build()`` runs this check on the assembly of the very compilation it ships and falls back
to the ``-DBSDFD_NO_ASYNC_LDS`` variant (ordinary compiler-managed LDS loads) when it fails.  ``tools/isa_mix.py
--check-async`` is the command-line front end.
"""
from __future__ import annotations

import collections
import re
from typing import Dict, List, Tuple


def kernel_body(lines: List[str], key: str) -> List[str]:
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and key in l and l.split(";")[0].rstrip().endswith(":"):
            start = i
            break
    if start is None:
        raise KeyError(f"kernel matching {key!r} not found")
    for j in range(start, len(lines)):
        if lines[j].strip().startswith("s_endpgm"):
            return lines[start:j + 1]
    return lines[start:]


def kernel_names(lines: List[str], fragment: str = "flow_kernel") -> List[str]:
    """Mangled names of every kernel in the file whose name contains ``fragment``."""
    out = []
    for l in lines:
        if l.startswith("_Z") and fragment in l and l.split(";")[0].rstrip().endswith(":"):
            out.append(l.split(":")[0].strip())
    return out


def loops(body: List[str]) -> List[Tuple[int, int]]:
    """(first, last) line indices of every label .. backward-branch pair."""
    labels, out = {}, []
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
        m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels:
            out.append((labels[m.group(1)], i))
    return out


def mix(body: List[str], lo: int, hi: int) -> collections.Counter:
    c = collections.Counter()
    for l in body[lo:hi + 1]:
        l = l.split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."):
            continue
        c[re.sub(r"_(e32|e64|sdwa|dpp)$", "", l.split()[0])] += 1
    return c


def euler_loop(body: List[str]):
    """The Euler-step loop = the SHORTEST loop that holds at least half of the MFMAs of the MFMA-richest loop (the tile loop
    around it holds the prologue's MFMAs as well).  None if the kernel has no loop with MFMAs."""
    cand = []
    for lo, hi in loops(body):
        c = mix(body, lo, hi)
        cand.append((sum(v for k, v in c.items() if k.startswith("v_mfma")), hi - lo, (lo, hi)))
    if not cand or max(n for n, _, _ in cand) == 0:
        return None
    top = max(n for n, _, _ in cand)
    return min((span, rng) for n, span, rng in cand if 2 * n >= top)[1]


_AGPR_BASE = 100000   # accumulation registers a0.. live in their own index space


def _regs(tok: str) -> set:
    """Vector registers named by an operand token: v12 -> {12}, v[4:7] -> {4..7}; AGPRs a3 / a[0:3] -> {_AGPR_BASE + i}."""
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return {int(m.group(2)) + (_AGPR_BASE if m.group(1) == "a" else 0)}
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        base = _AGPR_BASE if m.group(1) == "a" else 0
        return set(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
    return set()


def kernel_meta(lines: List[str], key: str) -> Dict[str, int]:
    meta = {}
    for i, l in enumerate(lines):
        if ".name:" in l and key in l:
            for l2 in lines[i:i + 16]:
                for f in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "private_segment_fixed_size"):
                    m = re.search(rf"\.{f}:\s+(\d+)", l2)
                    if m:
                        meta[f] = int(m.group(1))
            break
    return meta


def check_async_lines(lines: List[str], key: str) -> Tuple[int, List[str]]:
    """(number of asynchronous reads, violations) of one kernel.  A violation is
    * an instruction — spill stores and reloads included — that reads or writes a destination register of an asynchronous
      read before the next ``s_waitcnt lgkmcnt(0)``,
    * control flow with destinations still pending, or
    * spill traffic inside the Euler-step loop of a kernel that uses asynchronous reads."""
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
    if n_async:
        rng = euler_loop(body)
        if rng is not None:
            n_in = sum(1 for l in body[rng[0]:rng[1] + 1] if l.split(";")[0].strip().startswith("scratch_"))
            if n_in:
                bad.append(f"{n_in} scratch instructions inside the Euler-step loop")
    return n_async, bad


# Wait states (issued instructions; `s_nop N` counts N + 1) between an MFMA and a later non-MFMA instruction that READS its
# destination registers or, for a VALU instruction, OVERWRITES them: LLVM's GCNHazardRecognizer::checkMAIVALUHazards for gfx950 -
# XDL shapes (f16 / bf16 inputs) passes + 4, i.e. 8 for the 4-pass 16x16x32 and 12 for the 8-pass 32x32x16 (what hipcc pads to in
# straight-line code, and cdna_hip_programming.md §5.7: "8-pass XDL: 12 states"); fp32-input shapes passes + 2.  EVERY intervening
# instruction - another MFMA included - counts one state, a branch one state (a taken branch costs more in practice: the check errs
# on the safe side), an `s_nop N` N + 1.  An (empty) inline-asm statement counts what is inside it, i.e. possibly nothing - the
# compiler counts it as one, which is how a path can end up short.  The compiler inserts the padding - but hipcc 7.2 was caught
# (round 4) leaving ALL of it out on one path (an MFMA directly in front of a taken `s_cbranch`, its result read two instructions into
# the target block: the kernel computed with the stale accumulator) and ONE state short on the back edge of every run-time-depth
# layer loop (csrc/bsdfd.hip, BSDFD_LOOP_HEAD_PAD), so the build checks.  A load whose DESTINATION is the register is not a hazard
# (its data returns long after the matrix pipe has drained).
# opcode -> (wait states before a read, before a VALU overwrite)
MFMA_WAIT = {"v_mfma_f32_16x16x32_f16": (8, 8), "v_mfma_f32_16x16x32_bf16": (8, 8), "v_mfma_f32_16x16x16_f16": (8, 8),
             "v_mfma_f32_16x16x4_f32": (10, 10), "v_mfma_f32_32x32x16_f16": (12, 12), "v_mfma_f32_32x32x2_f32": (18, 18)}
MFMA_WAIT_DEFAULT = (20, 20)
_LOADS = ("ds_read", "global_load", "scratch_load", "buffer_load", "flat_load", "s_load", "s_buffer_load")


def _is_mfma(op: str) -> bool:
    return op.startswith("v_mfma") or op.startswith("v_smfmac")


def check_mfma_hazards_lines(lines: List[str], key: str) -> Tuple[int, List[str]]:
    """(number of MFMAs, violations): every non-MFMA instruction that reads (or, VALU, overwrites) a destination register of
    an MFMA must be the required number of wait states behind it on EVERY path (fall-through and branch targets are both
    followed).  MFMAs that take the result as an operand are exempt (the matrix pipe interlocks its own dependent issue)."""
    body = kernel_body(lines, key)
    ins, label_at = [], {}
    for raw in body:
        code = raw.split(";")[0].strip()
        if not code or code.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", raw)
            if m:
                label_at[m.group(1)] = len(ins)
            continue
        if code.endswith(":"):
            label_at[code[:-1]] = len(ins)
            continue
        toks = [t.strip(",") for t in code.split()]
        ins.append((toks[0], toks[1:], code))
    n_mfma, bad = 0, []
    for i, (op, toks, code) in enumerate(ins):
        if not _is_mfma(op):
            continue
        n_mfma += 1
        need_r, need_w = MFMA_WAIT.get(re.sub(r"_(e32|e64)$", "", op), MFMA_WAIT_DEFAULT)
        dst = _regs(toks[0]) if toks else set()
        if not dst:
            bad.append(f"`{code}`: destination operand not understood — cannot verify its wait states")
            continue
        # walk every path from the MFMA; `cur` = the destination registers that still hold ITS result (an instruction that
        # rewrites some of them takes those over: its own walk answers for them)
        stack, seen = [(i + 1, 0, frozenset(dst))], {}
        while stack:
            j, w, cur = stack.pop()
            while j < len(ins) and w < max(need_r, need_w) and cur:
                if seen.get((j, cur), 1 << 30) <= w:
                    break
                seen[(j, cur)] = w
                o, t, c = ins[j]
                if _is_mfma(o) and len(t) >= 4:
                    # a later MFMA: the matrix pipe interlocks an accumulator it takes over UNCHANGED (SrcC == the earlier vDst, the
                    # back-to-back accumulation these kernels use); a result consumed as SrcA / SrcB, or a partially overlapping
                    # SrcC, needs the software wait states like any other reader
                    srcs = set().union(*[_regs(x) for x in t[1:3]])
                    c_regs = _regs(t[3])
                    if c_regs != dst or cur != dst:
                        srcs |= c_regs
                    if cur & srcs and w < need_r:
                        bad.append(f"`{c}` reads the destination of `{code}` as a matrix operand after {w} wait states (needs {need_r})")
                        stack.clear()
                        break
                    cur = cur - _regs(t[0])      # (an accumulating MFMA rewrites its SrcC; either way the registers are its own now)
                elif t:
                    has_dst = o.startswith("v_") and not o.startswith("v_cmp") and not o.startswith("v_readlane") \
                        and not o.startswith("v_readfirstlane")
                    is_load = o.startswith(_LOADS)
                    writes = _regs(t[0]) if (has_dst or is_load) else set()
                    reads = set().union(*[_regs(x) for x in (t[1:] if (has_dst or is_load) else t)]) if t else set()
                    if o.startswith("v_writelane") or o.startswith("v_fmac") or o.startswith("v_pk_fmac"):
                        reads |= writes          # read-modify-write destinations
                    if o.startswith("v_permlane") and "swap" in o and len(t) > 1:
                        writes = writes | _regs(t[1])   # a swap reads and writes both operands
                        reads = reads | writes
                    if (cur & reads and w < need_r) or (cur & writes and has_dst and w < need_w):
                        kind = "reads" if cur & reads else "overwrites"
                        bad.append(f"`{c}` {kind} the destination of `{code}` after {w} wait states "
                                   f"(needs {need_r if kind == 'reads' else need_w})")
                        stack.clear()
                        break
                    cur = cur - writes           # redefined registers: later uses see the new value
                if o == "s_endpgm":
                    break
                m = re.search(r"(\.LBB\d+_\d+)", c) if o.startswith("s_cbranch") or o == "s_branch" else None
                if m and m.group(1) in label_at:
                    stack.append((label_at[m.group(1)], w + 1, cur))
                    if o == "s_branch":
                        break
                if o == "s_nop" and t:
                    w += int(t[0], 0) + 1
                else:
                    w += 1
                j += 1
    return n_mfma, bad


# ---- census: what a build MUST contain (the checks above report "no violations" for a kernel they do not find or cannot parse) ----
# Instantiation table of csrc/bsdfd.hip::kernel_ptr (mirrored here; tests/test_host_cpu.py holds the two in step): every
# (domain, width / 16, depths) x precision x {no Jacobian, Jacobian, Jacobian + fused sample/pdf}, precision f32 always with the
# run-time-depth loop.  Template arguments <DOMAIN, NM, PREC, JAC, NH, FUSED>.
_PREC_F32, _PREC_SPLIT3, _PREC_F16 = 1, 2, 3


def expected_flow_kernels(variant: str = "async") -> Dict[str, Dict[str, int]]:
    """Mangled-name fragment of every flow kernel the library is built from -> the least number of asynchronous `ds_read_b128`
    and of their wait statements its assembly must show (0 for the kernels that use none and for the `plain` fallback variant)."""
    out = {}
    for dom, shapes in ((0, ((2, (3, 0)), (4, (0,)))), (1, ((2, (4, 0)), (4, (6, 0))))):
        for nm, depths in shapes:
            for prec in (_PREC_F32, _PREC_SPLIT3, _PREC_F16):
                for nh in ((0,) if prec == _PREC_F32 else depths):
                    for jac, fused in ((0, 0), (1, 0), (1, 1)):
                        spec = {"async": 0, "waits": 0}
                        if variant == "async" and jac and nm == 2 and prec != _PREC_F32:
                            split = prec == _PREC_SPLIT3
                            if dom == 0 and nh == 3:      # block MIM: 25 (13) fragment reads, 5 (3) waits per step
                                spec = {"async": 25 if split else 13, "waits": 5 if split else 3}
                            elif dom == 1 and nh == 4:    # block MIMS: 21 (11) reads, 5 (4) waits
                                spec = {"async": 21 if split else 11, "waits": 5 if split else 4}
                        if prec == _PREC_F16 and not jac:
                            # packed-fp16 sigmoids (flow_dev.h: act_pack8): 2 x 4 destination-select writes per B fragment, NM / 2
                            # fragments per layer; the run-time-depth loop shows its body at least once
                            spec = dict(spec, sel=max(nh, 1) * (nm // 2) * 8)
                        out[f"flow_kernelILi{dom}ELi{nm}ELi{prec}ELb{jac}ELi{nh}ELb{fused}EE"] = spec
    for dom in (0, 1):                                    # csrc/flow32.hip, compiler-managed LDS reads
        for jac, fused, split in ((0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 0, 0)):   # <DOMAIN, JAC, FUSED, SPLIT>; (0, 0, 0): precision f16, samples-only
            # "sel": destination-select writes (the packed-fp16 sigmoids' SDWA halves, BSDFD_PK4_TRANS: 2 x 4 per fragment, one
            # fragment per 8 units) the forwarding-hazard check must see: 2 (3) hidden layers in front of the last one x 2 fragments
            out[f"flow_kernel32ILi{dom}ELb{jac}ELb{fused}ELb{split}EE"] = {"async": 0, "waits": 0, "sel": 0 if split else (32, 48)[dom]}
    out["flow_kernel32wE"] = {"async": 0, "waits": 0, "sel": 160}   # the 64 x 6 fp16 teacher on 32-query tiles: 5 layers x 4 fragments
    out["flow_kernel32cE"] = {"async": 0, "waits": 0}               # the 64 x 6 split3 net with the Jacobian on 32-query tiles (round 6)
    return out


def census_lines(lines: List[str], fragment: str = "flow_kernel") -> Dict[str, Dict[str, int]]:
    """Per kernel: MFMAs parsed, asynchronous reads and wait statements found (inside inline-asm brackets), and the .amdhsa
    metadata (an empty dict when the metadata block was not found)."""
    out = {}
    for k in kernel_names(lines, fragment):
        body = kernel_body(lines, k)
        n_mfma = n_async = n_wait = n_sel = 0
        in_asm = False
        for raw in body:
            l = raw.strip()
            if l.startswith(";;#ASMSTART") or l.startswith(";#ASMSTART"):
                in_asm = True
                continue
            if l.startswith(";;#ASMEND") or l.startswith(";#ASMEND"):
                in_asm = False
                continue
            code = l.split(";")[0].strip()
            if not code:
                continue
            op = code.split()[0]
            if _is_mfma(op):
                n_mfma += 1
            if in_asm and op == "ds_read_b128":
                n_async += 1
            if in_asm and op.startswith("s_waitcnt") and "lgkmcnt(0)" in code:
                n_wait += 1
            if in_asm and op.startswith("v_") and _dst_sel_write(op, code):
                n_sel += 1
        out[k] = {"mfma": n_mfma, "async": n_async, "waits": n_wait, "sel": n_sel, "meta": kernel_meta(lines, k)}
    return out


def verify_census(asm_paths: List[str], variant: str = "async", allow_scratch: bool = False, only: str = None) -> List[str]:
    """Problems that make the other checks of this module meaningless or the build unshippable: an expected flow kernel missing
    from the assembly (renamed labels, another mangling), an unexpected one, a kernel in which the parser sees no MFMA, fewer
    asynchronous reads / waits than the source issues (another mnemonic spelling), no metadata, or scratch memory in use.
    ``only``: restrict the expectation to the kernels whose name fragment starts with it ("flow_kernelI": csrc/bsdfd.hip,
    "flow_kernel32I": csrc/flow32.hip)."""
    found = {}
    for path in asm_paths:
        found.update(census_lines(open(path).read().splitlines()))
    expected = {f: v for f, v in expected_flow_kernels(variant).items() if only is None or f.startswith(only)}
    problems = []
    by_frag = {}
    for k, c in found.items():
        frag = next((f for f in expected if f in k), None)
        if frag is None and only is not None and only not in k:
            continue
        if frag is None:
            problems.append(f"unexpected flow kernel {k} (not in expected_flow_kernels: update the table with csrc/*.hip)")
        else:
            by_frag[frag] = c
    for frag, spec in expected.items():
        c = by_frag.get(frag)
        if c is None:
            problems.append(f"flow kernel {frag} not found in the assembly")
            continue
        if c["mfma"] < 1:
            problems.append(f"{frag}: no MFMA instruction recognised")
        if c["async"] < spec["async"] or c["waits"] < spec["waits"]:
            problems.append(f"{frag}: {c['async']} asynchronous ds_read_b128 / {c['waits']} waits recognised, the source issues "
                            f"{spec['async']} / {spec['waits']}")
        if c.get("sel", 0) < spec.get("sel", 0):
            problems.append(f"{frag}: {c.get('sel', 0)} destination-select writes recognised in its inline asm, the source issues {spec['sel']}")
        if not c["meta"] or "vgpr_count" not in c["meta"]:
            problems.append(f"{frag}: .amdhsa metadata not found")
        elif not allow_scratch and c["meta"].get("private_segment_fixed_size", 0) != 0:
            problems.append(f"{frag}: {c['meta']['private_segment_fixed_size']} B of scratch per lane (spills)")
    return problems


def check_file(path: str, fragment: str = "flow_kernel") -> Dict[str, Tuple[int, List[str]]]:
    """Every kernel of the assembly file whose name contains ``fragment`` -> (asynchronous reads, violations of the
    asynchronous-read discipline)."""
    lines = open(path).read().splitlines()
    return {k: check_async_lines(lines, k) for k in kernel_names(lines, fragment)}


def check_file_mfma(path: str, fragment: str = "flow_kernel") -> Dict[str, Tuple[int, List[str]]]:
    """Every kernel of the assembly file whose name contains ``fragment`` -> (MFMAs, missing-wait-state violations)."""
    lines = open(path).read().splitlines()
    return {k: check_mfma_hazards_lines(lines, k) for k in kernel_names(lines, fragment)}


# VALU write of a VGPR -> v_permlane{16,32}_swap touching it: 2 wait states (LLVM's gfx950 rule, cdna_hip_programming.md §4:
# "VALU write of either swap operand must be followed by 2 wait states").  The compiler pads these too; checked for the same
# reason as the MFMA results above (the reductions of the Jacobian run through these swaps).
SWAP_WAIT = 2


def check_swap_hazards_lines(lines: List[str], key: str) -> Tuple[int, List[str]]:
    """(number of v_permlane*_swap instructions, violations): every VALU instruction that writes a register a later
    v_permlane16_swap / v_permlane32_swap reads or exchanges must be SWAP_WAIT wait states ahead of it on every path."""
    body = kernel_body(lines, key)
    ins, label_at = [], {}
    for raw in body:
        code = raw.split(";")[0].strip()
        if not code or code.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", code)
            if m:
                label_at[m.group(1)] = len(ins)
            continue
        if code.endswith(":"):
            label_at[code[:-1]] = len(ins)
            continue
        toks = [t.strip(",") for t in code.split()]
        ins.append((toks[0], toks[1:], code))
    n_swap = sum(1 for o, _, _ in ins if o.startswith("v_permlane") and "swap" in o)
    bad = []
    for i, (op, toks, code) in enumerate(ins):
        if not op.startswith("v_") or op.startswith("v_cmp") or op.startswith("v_readlane") or op.startswith("v_readfirstlane") \
                or op.startswith("v_mfma") or op.startswith("v_smfmac") or not toks:
            continue
        dst = _regs(toks[0])
        if op.startswith("v_permlane") and "swap" in op and len(toks) > 1:
            dst = dst | _regs(toks[1])   # a swap writes both operands
        if not dst:
            continue
        stack, seen = [(i + 1, 0)], {}
        while stack:
            j, w = stack.pop()
            while j < len(ins) and w < SWAP_WAIT:
                if seen.get(j, 1 << 30) <= w:
                    break
                seen[j] = w
                o, t, c = ins[j]
                if o.startswith("v_permlane") and "swap" in o and len(t) > 1 and dst & (_regs(t[0]) | _regs(t[1])):
                    bad.append(f"`{c}` exchanges a register `{code}` wrote {w} wait states earlier (needs {SWAP_WAIT})")
                    stack.clear()
                    break
                if o == "s_endpgm":
                    break
                m = re.search(r"(\.LBB\d+_\d+)", c) if o.startswith("s_cbranch") or o == "s_branch" else None
                if m and m.group(1) in label_at:
                    stack.append((label_at[m.group(1)], w + 1))
                    if o == "s_branch":
                        break
                w += int(t[0], 0) + 1 if (o == "s_nop" and t) else 1
                j += 1
    return n_swap, bad


def check_file_swap(path: str, fragment: str = "flow_kernel") -> Dict[str, Tuple[int, List[str]]]:
    lines = open(path).read().splitlines()
    return {k: check_swap_hazards_lines(lines, k) for k in kernel_names(lines, fragment)}


# Forwarding hazards of gfx940-class parts (LLVM GCNHazardRecognizer::checkVALUHazards, hasTransForwardingHazard /
# hasDstSelForwardingHazard): a non-transcendental VALU instruction that reads the result of a transcendental, and ANY VALU
# instruction that reads a register written through a destination select (SDWA dst_sel other than DWORD, op_sel to the high
# half), must be at least one wait state behind it.  The compiler pads its own code; the packed-fp16 sigmoids of csrc/flow32.hip
# (BSDFD_PK4_TRANS) are inline asm, which its hazard recogniser does not look into — so the build checks every kernel.
FORWARD_WAIT = 1
_TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def _is_trans(op: str) -> bool:
    return op.startswith(_TRANS)


def _dst_sel_write(op: str, code: str) -> bool:
    if op.endswith("_sdwa"):
        m = re.search(r"dst_sel:(\w+)", code)
        return bool(m) and m.group(1) != "DWORD"
    m = re.search(r"op_sel:\[([01,\s]+)\]", code)      # VOP3 / VOP3P op_sel: the LAST bit is the destination's
    return bool(m) and op.startswith("v_") and not op.startswith("v_pk_") and m.group(1).replace(" ", "").split(",")[-1] == "1"


def check_forwarding_hazards_lines(lines: List[str], key: str) -> Tuple[int, List[str]]:
    """(number of destination-select writes, violations) of one kernel: see FORWARD_WAIT."""
    body = kernel_body(lines, key)
    ins, label_at = [], {}
    for raw in body:
        code = raw.split(";")[0].strip()
        if not code or code.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", code)
            if m:
                label_at[m.group(1)] = len(ins)
            continue
        if code.endswith(":"):
            label_at[code[:-1]] = len(ins)
            continue
        toks = [t.strip(",") for t in code.split()]
        ins.append((toks[0], toks[1:], code))
    n_sel, bad = 0, []
    for i, (op, toks, code) in enumerate(ins):
        if not op.startswith("v_") or not toks:
            continue
        trans, sel = _is_trans(op), _dst_sel_write(op, code)
        n_sel += sel
        if not (trans or sel):
            continue
        dst = _regs(toks[0])
        if not dst:
            bad.append(f"`{code}`: destination operand not understood — cannot verify its wait state")
            continue
        # the instruction(s) within FORWARD_WAIT wait states on every path
        stack = [(i + 1, 0)]
        while stack:
            j, w = stack.pop()
            while j < len(ins) and w < FORWARD_WAIT:
                o, t, c = ins[j]
                if o.startswith("v_") and t:
                    has_dst = not o.startswith("v_cmp") and not o.startswith("v_readlane") and not o.startswith("v_readfirstlane")
                    reads = set().union(*[_regs(x) for x in (t[1:] if has_dst else t)])
                    if has_dst and (o.startswith("v_fmac") or o.startswith("v_pk_fmac") or "UNUSED_PRESERVE" in c
                                    or (o.endswith("_f16") and not o.startswith("v_pk_") and not o.startswith("v_cvt_pk"))):
                        reads |= _regs(t[0])     # read-modify-write destinations (16-bit results keep the other half)
                    if dst & reads and (sel or not _is_trans(o)):
                        what = "a destination-select write" if sel else "a transcendental"
                        bad.append(f"`{c}` reads the result of `{code}` ({what}) {w} wait states behind it (needs {FORWARD_WAIT})")
                        stack.clear()
                        break
                if o == "s_endpgm":
                    break
                m = re.search(r"(\.LBB\d+_\d+)", c) if o.startswith("s_cbranch") or o == "s_branch" else None
                if m and m.group(1) in label_at:
                    stack.append((label_at[m.group(1)], w + 1))
                    if o == "s_branch":
                        break
                w += int(t[0], 0) + 1 if (o == "s_nop" and t) else 1
                j += 1
    return n_sel, bad


def check_file_forwarding(path: str, fragment: str = "flow_kernel") -> Dict[str, Tuple[int, List[str]]]:
    lines = open(path).read().splitlines()
    return {k: check_forwarding_hazards_lines(lines, k) for k in kernel_names(lines, fragment)}
