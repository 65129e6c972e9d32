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


def _regs(tok: str) -> set:
    """VGPR indices named by an operand token: v12 -> {12}, v[4:7] -> {4..7}."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
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
            continue
        stack, seen = [(i + 1, 0)], {}
        while stack:
            j, w = stack.pop()
            while j < len(ins) and w < max(need_r, need_w):
                if seen.get(j, 1 << 30) <= w:
                    break
                seen[j] = w
                o, t, c = ins[j]
                if not _is_mfma(o) and t:
                    has_dst = o.startswith("v_") and not o.startswith("v_cmp") and not o.startswith("v_readlane") \
                        and not o.startswith("v_readfirstlane")
                    is_load = o.startswith(_LOADS)
                    writes = _regs(t[0]) if (has_dst or is_load) else set()
                    reads = set().union(*[_regs(x) for x in (t[1:] if (has_dst or is_load) else t)]) if t else set()
                    if o.startswith("v_writelane") or o.startswith("v_fmac") or o.startswith("v_pk_fmac"):
                        reads |= writes          # read-modify-write destinations
                    if (dst & reads and w < need_r) or (dst & writes and has_dst and w < need_w):
                        kind = "reads" if dst & reads else "overwrites"
                        bad.append(f"`{c}` {kind} the destination of `{code}` after {w} wait states "
                                   f"(needs {need_r if kind == 'reads' else need_w})")
                        stack.clear()
                        break
                    if dst <= writes and not (dst & reads):
                        break                    # fully redefined: later uses see the new value
                if o == "s_endpgm":
                    break
                m = re.search(r"(\.LBB\d+_\d+)", c) if o.startswith("s_cbranch") or o == "s_branch" else None
                if m and m.group(1) in label_at:
                    stack.append((label_at[m.group(1)], w + 1))
                    if o == "s_branch":
                        break
                if o == "s_nop" and t:
                    w += int(t[0], 0) + 1
                else:
                    w += 1
                j += 1
    return n_mfma, bad


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
