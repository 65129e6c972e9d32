"""ctypes binding of the C-ABI shared library ``libbsdfd.so`` (include/bsdfd.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There
is NO fallback: if the library is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
DEFAULT_LIB_PATH = os.path.join(_HERE, "libbsdfd.so")
LIB_PATH = os.environ.get("BSDFD_LIB_PATH") or DEFAULT_LIB_PATH  # override: A/B builds
SRC_PATH = os.path.join(_HERE, "csrc", "bsdfd.hip")
SRC32_PATH = os.path.join(_HERE, "csrc", "flow32.hip")   # the 32-query-tile flow kernels
SRC_PATHS = [SRC_PATH, SRC32_PATH, os.path.join(_HERE, "csrc", "wavefront.hip"), os.path.join(_HERE, "csrc", "encoding.hip"),
             os.path.join(_HERE, "csrc", "measured.hip"), os.path.join(_HERE, "csrc", "bucket.hip"),
             os.path.join(_HERE, "csrc", "clock.hip")]  # translation units of libbsdfd.so
FLOW_TUS = (SRC_PATH, SRC32_PATH)   # ... whose device assembly the build verifies (_asmcheck)
DEP_PATHS = SRC_PATHS + [os.path.join(_HERE, "csrc", f) for f in ("common.h", "flow_dev.h", "flow32.h")]
INCLUDE_DIR = os.path.join(ROOT, "include")
ASM_CACHE_DIR = os.path.join(ROOT, "build", "asm")   # device assembly of the last product build (bsdfd.s, flow32.s)

PREC_DEFAULT, PREC_F32, PREC_SPLIT3, PREC_F16 = 0, 1, 2, 3
PRECISIONS = {"default": PREC_DEFAULT, "f32": PREC_F32, "split3": PREC_SPLIT3, "f16": PREC_F16}
PLUGIN_MEASURED, PLUGIN_FULLSPHERE = 0, 1

# every symbol include/bsdfd.h declares
EXPORTS = (
    "bsdfd_create", "bsdfd_create_from_file", "bsdfd_destroy", "bsdfd_get_info", "bsdfd_get_tile",
    "bsdfd_flops_per_query", "bsdfd_network_sampling", "bsdfd_network_pdf",
    "bsdfd_plugin_sample", "bsdfd_plugin_pdf", "bsdfd_plugin_sample_pdf", "bsdfd_plugin_sample_multi", "bsdfd_plugin_pdf_multi",
    "bsdfd_plugin_sample_pdf_multi",
    "bsdfd_context_bytes", "bsdfd_plugin_sample_ex", "bsdfd_plugin_pdf_ex", "bsdfd_plugin_sample_multi_ex",
    "bsdfd_plugin_pdf_multi_ex", "bsdfd_plugin_sample_pdf_multi_ex",
    "bsdfd_flow_samples_only", "bsdfd_wf_primary", "bsdfd_wf_shade",
    "bsdfd_positional_encoding", "bsdfd_bucket_workspace_bytes", "bsdfd_bucket_by_material",
    "bsdfd_gather_lanes", "bsdfd_scatter_lanes",
    "bsdfd_measured_create_from_file", "bsdfd_measured_destroy", "bsdfd_measured_get_info", "bsdfd_measured_eval",
    "bsdfd_measured_sample_weight",
    "bsdfd_set_profiling", "bsdfd_profile_read", "bsdfd_profile_read_op", "bsdfd_profile_clock_mhz", "bsdfd_last_kernel_ms", "bsdfd_shader_clock_mhz",
    "bsdfd_last_error", "bsdfd_version", "bsdfd_abi_version",
)
ABI_VERSION = 6   # BSDFD_ABI_VERSION of include/bsdfd.h these ctypes structs mirror (checked against the library in lib())


class WfScene(C.Structure):
    """bsdfd_wf_scene (include/bsdfd.h)."""
    _fields_ = [("cam_origin", C.c_float * 3), ("cam_right", C.c_float * 3), ("cam_up", C.c_float * 3),
                ("cam_forward", C.c_float * 3), ("tan_half_fov", C.c_float), ("width", C.c_int32),
                ("height", C.c_int32), ("sphere_center", C.c_float * 3), ("sphere_radius", C.c_float),
                ("albedo", C.c_float * 3), ("env_width", C.c_int32), ("env_height", C.c_int32),
                ("n_extra_spheres", C.c_int32), ("extra_spheres", (C.c_float * 4) * 31), ("has_plane", C.c_int32),
                ("plane_y", C.c_float), ("checker_scale", C.c_float), ("checker_color0", C.c_float),
                ("checker_color1", C.c_float)]


class Opts(C.Structure):
    """bsdfd_opts (include/bsdfd.h): optional arguments of the plugin-level *_ex calls."""
    _fields_ = [("ctx_out", C.c_void_p), ("ctx_in", C.c_void_p), ("rng_index", C.c_void_p), ("row_index", C.c_void_p)]


def opts(ctx_out=None, ctx_in=None, rng_index=None, byte_offset_rng: int = 0, row_index=None, byte_offset_row: int = 0):
    """A bsdfd_opts from torch tensors (None = absent); ``byte_offset_rng`` / ``byte_offset_row`` advance the index pointers."""
    o = Opts()
    o.ctx_out = None if ctx_out is None else ctx_out.data_ptr()
    o.ctx_in = None if ctx_in is None else ctx_in.data_ptr()
    o.rng_index = None if rng_index is None else rng_index.data_ptr() + byte_offset_rng
    o.row_index = None if row_index is None else row_index.data_ptr() + byte_offset_row
    return o


class Desc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("domain", "width", "n_hidden", "pe_bands", "base_hidden",
                                         "base_pe_bands", "precision", "tile")] + \
               [(n, C.POINTER(C.c_float)) for n in ("w_in", "w_hidden", "w_out", "base_w1", "base_b1",
                                                    "base_w2", "base_b2")]


KERNEL_SOURCES = (SRC_PATH, SRC32_PATH, os.path.join(_HERE, "csrc", "flow_dev.h"))


def kernel_source_sha256() -> str:
    """Fingerprint of the flow kernels' source (csrc/bsdfd.hip, csrc/flow32.hip, csrc/flow_dev.h): the committed profile summaries
    (HBM traffic from the PMC passes, the instruction-issue model from the ISA) record it; bench.py withholds them and a CPU test
    fails when the kernels have changed since."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(f, "rb").read())
    return h.hexdigest()


HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed",
               "-Wno-unused-command-line-argument"]


def _check_asm(asm_path: str):
    """Violations of the asynchronous-LDS-read discipline in the assembly of a compilation of csrc/bsdfd.hip
    (``_asmcheck.check_file``): {kernel: [messages]} for the kernels that have any.  A seam of its own so that a test can
    feed the build a doctored result."""
    from . import _asmcheck
    return {k: bad for k, (n, bad) in _asmcheck.check_file(asm_path).items() if bad}


def _check_asm_mfma(asm_path: str):
    """Instructions that read or overwrite an MFMA's destination before the wait states the ISA requires
    (``_asmcheck.check_file_mfma``): {kernel: [messages]}.  The compiler is supposed to pad for these; ROCm 7.2 was caught
    leaving the padding out behind a taken branch (csrc/bsdfd.hip, the base-net MFMAs), silently computing with a stale
    accumulator — there is no fallback for this class, the build refuses to ship."""
    from . import _asmcheck
    out = {k: bad for k, (n, bad) in _asmcheck.check_file_mfma(asm_path).items() if bad}
    # ... and the second class of software-managed hazards these kernels rely on: a VALU write followed by a v_permlane*_swap of
    # the same register (2 wait states; the Jacobian's lane reductions run through these swaps)
    for k, (n, bad) in _asmcheck.check_file_swap(asm_path).items():
        if bad:
            out.setdefault(k, []).extend(bad)
    # ... and the third: results of transcendentals and of destination-select writes read one instruction later (the packed-fp16
    # sigmoids of csrc/flow32.hip are inline asm, invisible to the compiler's hazard recogniser)
    for k, (n, bad) in _asmcheck.check_file_forwarding(asm_path).items():
        if bad:
            out.setdefault(k, []).extend(bad)
    return out


def _flow_tu_cmd(td: str, src: str, extra):
    stem = os.path.basename(src)[:-4]
    return ["hipcc", *HIPCC_FLAGS, *extra, "-save-temps=obj", "-I", INCLUDE_DIR, "-c", src, "-o", os.path.join(td, stem + ".o")]


def _flow_tu_asm(td: str, src: str) -> str:
    import glob
    stem = os.path.basename(src)[:-4]
    asm = [f for f in glob.glob(os.path.join(td, stem + "*.s")) if "gfx950" in os.path.basename(f)]
    if len(asm) != 1:
        raise RuntimeError(f"expected one gfx950 assembly file from -save-temps, found {asm}")
    return asm[0]


def _compile_flow_tu(td: str, extra, verbose: bool, src: str = None):
    """Compile a flow-kernel translation unit (default csrc/bsdfd.hip) to td/<stem>.o keeping the device assembly of THIS
    compilation (-save-temps=obj); returns the path of that assembly."""
    import glob
    src = src or SRC_PATH
    for f in glob.glob(os.path.join(td, os.path.basename(src)[:-4] + "*")):
        os.remove(f)
    cmd = _flow_tu_cmd(td, src, extra)
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=td)
    return _flow_tu_asm(td, src)


def _census(asm_paths, variant: str, only: str = None):
    """``_asmcheck.verify_census`` — a seam of its own so that a test can feed the build doctored assembly."""
    from . import _asmcheck
    return _asmcheck.verify_census(asm_paths, variant, only=only)


def build(force: bool = False, verbose: bool = False, lib_path: str = None) -> str:
    """Compile libbsdfd.so for gfx950 with hipcc (cross-compiles without a GPU).

    The 16-query-tile flow kernels read their weight fragments with inline-asm LDS loads whose safety depends on the toolchain's
    register allocation (csrc/bsdfd.hip, lds_read_b128_async_at), so the build verifies the assembly of its own compilation
    (``_asmcheck``) and, if that fails — a violation, OR the checker not recognising the reads it is meant to verify — REBUILDS
    the kernels with ``-DBSDFD_NO_ASYNC_LDS`` (compiler-managed LDS loads, ~2 % slower) instead of shipping a library that could
    read stale weights.  The assembly of both flow-kernel translation units is then held to

    * a CENSUS (``_asmcheck.verify_census``): every expected instantiation present, MFMAs and metadata parsed in each, the known
      number of asynchronous reads and waits recognised, no scratch memory — so that a toolchain which changes label, mnemonic
      or metadata syntax cannot turn the checks below into "0 violations of 0 instructions";
    * the MFMA / lane-swap wait-state check (``_check_asm_mfma``).

    Either failing aborts the build: the library could compute with stale registers.  ``BSDFD_ALLOW_UNVERIFIED_BUILD=1`` ships it
    anyway, loudly — ``<lib>.build.json`` ("unverified": true), ``<lib>.asmcheck.txt`` (every finding) and ``bsdfd_version()``
    ("UNVERIFIED BUILD") record it.  ``<lib>.build.json`` and ``bsdfd_version()`` also say which LDS variant shipped.

    ``BSDFD_COMPILER_ONLY_BUILD=1`` — for a toolchain whose assembly the parser does not understand: both
    flow-kernel translation units are compiled with ``-DBSDFD_NO_ASYNC_LDS -DBSDFD_NO_SDWA_PACK``, i.e. WITHOUT the two constructs
    whose safety rests on the assembly checks (the inline-asm LDS reads; the packed-fp16 sigmoids' SDWA blocks): every LDS read and
    every fp16 sigmoid is then ordinary compiler-scheduled code, a few per cent slower, same results.  The assembly is not parsed at
    all (only searched, as text, for the two constructs); the hand-padded wait states behind the MFMAs the compiler was caught
    omitting stay in the source.  Recorded in ``<lib>.build.json`` ("compiler_only": true) and in ``bsdfd_version()``."""
    import json
    import tempfile
    out = lib_path or LIB_PATH
    hdr = os.path.join(INCLUDE_DIR, "bsdfd.h")
    deps = DEP_PATHS + [os.path.join(_HERE, "_asmcheck.py")]
    if (not force and os.path.exists(out)
            and os.path.getmtime(out) >= max([os.path.getmtime(hdr)] + [os.path.getmtime(f) for f in deps])):
        return out
    # compile to a temporary name and rename into place: a concurrent process (another rank of a torchrun
    # launch, a parallel test worker) never dlopens a half-written library
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    tmp = f"{out}.tmp.{os.getpid()}"
    info = {"variant": "async", "violations": {}, "hipcc": None, "unverified": False, "compiler_only": False}
    allow_unverified = os.environ.get("BSDFD_ALLOW_UNVERIFIED_BUILD") == "1"
    compiler_only = os.environ.get("BSDFD_COMPILER_ONLY_BUILD") == "1"
    CO_FLAGS = ["-DBSDFD_NO_ASYNC_LDS", "-DBSDFD_NO_SDWA_PACK", "-DBSDFD_COMPILER_ONLY_BUILD"]
    try:
        v = subprocess.run(["hipcc", "--version"], capture_output=True, text=True)
        info["hipcc"] = next((l.strip() for l in v.stdout.splitlines() if "version" in l.lower()), None)
    except OSError:
        pass
    with tempfile.TemporaryDirectory(prefix="bsdfd_build_") as td:
        side, procs = [], []
        for src in SRC_PATHS[1:]:   # the other translation units in parallel with csrc/bsdfd.hip (the long one)
            obj = os.path.join(td, os.path.basename(src)[:-4] + ".o")
            side.append(obj)
            cmd = (_flow_tu_cmd(td, src, CO_FLAGS if compiler_only else []) if src in FLOW_TUS
                   else ["hipcc", *HIPCC_FLAGS, "-I", INCLUDE_DIR, "-c", src, "-o", obj])
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd, cwd=td)))

        def reap(kill: bool):
            """Wait for (or kill) every child; the first failed command, or None.  Never raises: it also runs while another
            exception is in flight, which it must not replace."""
            failed = None
            for cmd, pr in procs:
                if kill and pr.poll() is None:
                    pr.kill()
                if pr.wait() != 0 and failed is None and not kill:
                    failed = (pr.returncode, cmd)
            return failed

        findings = []
        try:
            asm = _compile_flow_tu(td, CO_FLAGS if compiler_only else [], verbose)
            bad = {} if compiler_only else _check_asm(asm)
            blind = [] if compiler_only else _census([asm], "async", only="flow_kernelI")
            if compiler_only:
                info["variant"], info["compiler_only"] = "plain", True
                print("=" * 100 + "\nbsdfd build: BSDFD_COMPILER_ONLY_BUILD=1 — no inline-asm LDS reads, no SDWA sigmoids; the device assembly is NOT "
                      "inspected (bsdfd_version() says so)\n" + "=" * 100, flush=True)
            if bad or blind:
                info["variant"] = "plain"
                info["violations"] = {k: v[:8] for k, v in bad.items()} or {"census": blind[:8]}
                print("=" * 100 + "\nbsdfd build: THE ASYNCHRONOUS LDS READS OF csrc/bsdfd.hip CANNOT BE SHIPPED WITH THIS TOOLCHAIN:", flush=True)
                for k, v in bad.items():
                    print(f"  {k}: {len(v)} instruction(s) touch destination registers before their wait, e.g. {v[0]}", flush=True)
                for msg in blind[:8]:
                    print(f"  census: {msg}", flush=True)
                print("rebuilding the flow kernels with -DBSDFD_NO_ASYNC_LDS (compiler-managed LDS loads, ~2 % slower)\n" + "=" * 100,
                      flush=True)
                findings += [f"async: {k}: {m}" for k, v in bad.items() for m in v] + [f"async census: {m}" for m in blind]
                asm = _compile_flow_tu(td, ["-DBSDFD_NO_ASYNC_LDS"], verbose)
                still = _check_asm(asm)
                if still:
                    raise RuntimeError(f"the fallback build still fails the assembly check: {still}")
        except BaseException:
            reap(kill=True)
            raise
        failed = reap(kill=False)
        if failed:
            raise subprocess.CalledProcessError(*failed)
        asms = [asm] + [_flow_tu_asm(td, src) for src in FLOW_TUS[1:]]
        fatal = []
        if compiler_only:
            # a TEXT search, independent of the parser: neither of the two constructs may be left in the code
            text = "".join(open(a).read() for a in asms)
            blocks = [blk.split("#ASMEND")[0] for blk in text.split("#ASMSTART")[1:]]   # (the compiler's own SDWA forms are its business)
            info["inline_asm_constructs_left"] = {"ds_read_b128 inside inline asm": sum(1 for blk in blocks if "ds_read_b128" in blk),
                                                  "_sdwa inside inline asm": sum(1 for blk in blocks if "_sdwa" in blk)}
            if any(info["inline_asm_constructs_left"].values()):
                raise RuntimeError(f"compiler-only build still contains inline-asm constructs: {info['inline_asm_constructs_left']}")
        else:
            fatal = [f"census: {m}" for m in _census(asms, info["variant"])]
            for a in asms:
                for k, msgs in _check_asm_mfma(a).items():
                    fatal += [f"{os.path.basename(a)}: {k}: {m}" for m in msgs]
        findings += fatal
        if findings:
            with open(out + ".asmcheck.txt", "w") as f:
                f.write("\n".join(findings) + "\n")
        elif os.path.exists(out + ".asmcheck.txt"):
            os.remove(out + ".asmcheck.txt")
        if fatal:
            msg = ("bsdfd build: the device assembly of this toolchain's compilation fails verification "
                   f"({len(fatal)} finding(s), all of them in {out}.asmcheck.txt) — e.g. {fatal[0]}.  Either the checker no longer "
                   "recognises what it must verify (census) or MFMA results / swapped lanes are consumed before the wait states the "
                   "ISA requires: the library could compute with stale registers "
                   "(bsdf_diffusion_sampling_amd/_asmcheck.py: verify_census, check_mfma_hazards_lines, check_swap_hazards_lines).")
            if not allow_unverified:
                raise RuntimeError(msg + "  The build is refusing to ship it; BSDFD_ALLOW_UNVERIFIED_BUILD=1 overrides (recorded in the "
                                         "build info and in bsdfd_version()).")
            print("=" * 100 + "\n" + msg + "\nBSDFD_ALLOW_UNVERIFIED_BUILD=1: SHIPPING IT ANYWAY, marked UNVERIFIED\n" + "=" * 100, flush=True)
            info["unverified"] = True
            extra = ["-DBSDFD_UNVERIFIED_BUILD"] + (["-DBSDFD_NO_ASYNC_LDS"] if info["variant"] == "plain" else [])
            _compile_flow_tu(td, extra, verbose)   # the marker goes into bsdfd_version()
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(td, "bsdfd.o"), *side, "-o", tmp]
        if verbose:
            print(" ".join(cmd), flush=True)
        try:
            subprocess.run(cmd, check=True)
            # the build info first (atomically), then the library: a concurrent rank never sees a new library with stale info
            with open(f"{out}.build.json.tmp.{os.getpid()}", "w") as f:
                json.dump(info, f, indent=1)
            os.replace(f"{out}.build.json.tmp.{os.getpid()}", out + ".build.json")
            os.replace(tmp, out)
            if os.path.abspath(out) == DEFAULT_LIB_PATH and not fatal and not compiler_only:
                # the verified assembly of the library that is NOW in place, kept for tools/isa_mix.py and the census tests (build/ is
                # scratch; a read-only tree must not fail the build).  Only for the default product path — never for an A/B library
                # selected with $BSDFD_LIB_PATH — and stamped with the kernel sources' hash, which the readers compare.
                try:
                    import shutil
                    os.makedirs(ASM_CACHE_DIR, exist_ok=True)
                    for a in asms:
                        shutil.copyfile(a, os.path.join(ASM_CACHE_DIR, os.path.basename(a).split("-hip-")[0] + ".s"))
                    with open(os.path.join(ASM_CACHE_DIR, "source.sha256"), "w") as f:
                        f.write(kernel_source_sha256() + "\n")
                except OSError as exc:
                    print(f"bsdfd build: could not cache the verified assembly under {ASM_CACHE_DIR}: {exc}", flush=True)
        finally:
            for leftover in (tmp, f"{out}.build.json.tmp.{os.getpid()}"):
                if os.path.exists(leftover):
                    os.remove(leftover)
    if verbose:
        print(f"bsdfd build: shipped the {'asynchronous-LDS' if info['variant'] == 'async' else 'FALLBACK (compiler-managed LDS)'} "
              f"variant of the flow kernels{' — COMPILER-ONLY build' if compiler_only else ''} ({out})", flush=True)
    return out


def cached_asm_paths():
    """[bsdfd.s, flow32.s] under build/asm/ if that cache is the assembly of the CURRENT kernel sources (hash stamp written by
    ``build()``), else None — the readers (tools/isa_mix.py, the census tests) rebuild instead of trusting file times."""
    paths = [os.path.join(ASM_CACHE_DIR, f) for f in ("bsdfd.s", "flow32.s")]
    try:
        stamp = open(os.path.join(ASM_CACHE_DIR, "source.sha256")).read().strip()
    except OSError:
        return None
    return paths if stamp == kernel_source_sha256() and all(os.path.exists(q) for q in paths) else None


def build_info(lib_path: str = None) -> dict:
    """What the last ``build()`` of this library shipped: {"variant": "async" | "plain", "violations": {...}, "hipcc": ...}
    (empty when the library was built some other way)."""
    import json
    try:
        return json.load(open((lib_path or LIB_PATH) + ".build.json"))
    except (OSError, ValueError):
        return {}


_lib = None


def lib():
    """Load the library (raises if it has not been built — no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # Load PyTorch's bundled HIP runtime FIRST: libbsdfd.so needs libamdhip64.so.7 and must bind to
    # the same runtime instance torch uses (two HIP runtimes in one process do not share devices,
    # streams or allocations: the second one reports "no ROCm-capable device").
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, fp = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_void_p
    L.bsdfd_create.argtypes = [C.POINTER(Desc), C.POINTER(vp)]
    L.bsdfd_create_from_file.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
    L.bsdfd_destroy.argtypes = [vp]
    L.bsdfd_destroy.restype = None
    L.bsdfd_get_info.argtypes = [vp] + [C.POINTER(i32)] * 4
    L.bsdfd_get_tile.argtypes = [vp, i32, C.POINTER(i32)]
    L.bsdfd_flops_per_query.argtypes = [vp, i32]
    L.bsdfd_flops_per_query.restype = i64
    L.bsdfd_network_sampling.argtypes = [vp, fp, fp, u64, u64, i64, i32, fp, fp, vp]
    L.bsdfd_network_pdf.argtypes = [vp, fp, fp, i64, i32, fp, vp]
    L.bsdfd_plugin_sample.argtypes = [vp, i32, fp, fp, u64, u64, i64, i32, fp, fp, vp]
    L.bsdfd_plugin_pdf.argtypes = [vp, i32, fp, fp, i64, i32, fp, vp]
    L.bsdfd_plugin_sample_pdf.argtypes = [vp, i32, fp, fp, fp, u64, u64, i64, i32, fp, fp, fp, vp]
    L.bsdfd_plugin_sample_multi.argtypes = [C.POINTER(vp), i32, C.POINTER(i64), i32, fp, fp, u64, u64, i32, fp, fp, vp]
    L.bsdfd_plugin_sample_pdf_multi.argtypes = [C.POINTER(vp), i32, C.POINTER(i64), i32, fp, fp, fp, u64, u64, i32, fp, fp, fp, vp]
    L.bsdfd_plugin_pdf_multi.argtypes = [C.POINTER(vp), i32, C.POINTER(i64), i32, fp, fp, i32, fp, vp]
    L.bsdfd_context_bytes.argtypes = [vp, i64, i32]
    L.bsdfd_context_bytes.restype = i64
    op = C.POINTER(Opts)
    L.bsdfd_plugin_sample_ex.argtypes = [vp, i32, fp, fp, u64, u64, i64, i32, fp, fp, op, vp]
    L.bsdfd_plugin_pdf_ex.argtypes = [vp, i32, fp, fp, i64, i32, fp, op, vp]
    L.bsdfd_plugin_sample_multi_ex.argtypes = [C.POINTER(vp), i32, C.POINTER(i64), i32, fp, fp, u64, u64, i32, fp, fp, op, vp]
    L.bsdfd_plugin_pdf_multi_ex.argtypes = [C.POINTER(vp), i32, C.POINTER(i64), i32, fp, fp, i32, fp, op, vp]
    L.bsdfd_plugin_sample_pdf_multi_ex.argtypes = [C.POINTER(vp), i32, C.POINTER(i64), i32, fp, fp, fp, u64, u64, i32, fp, fp, fp, op, vp]
    L.bsdfd_flow_samples_only.argtypes = [vp, fp, fp, i64, i32, fp, vp]
    L.bsdfd_wf_primary.argtypes = [C.POINTER(WfScene), i32, i32, i32, u64, u64, fp, fp, fp, fp, fp, vp]
    L.bsdfd_wf_shade.argtypes = [C.POINTER(WfScene), fp, i32, i32, i32, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, vp]
    L.bsdfd_positional_encoding.argtypes = [fp, i64, i32, i32, i32, i32, fp, vp]
    L.bsdfd_measured_create_from_file.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.bsdfd_measured_destroy.argtypes = [vp]
    L.bsdfd_measured_destroy.restype = None
    L.bsdfd_measured_get_info.argtypes = [vp] + [C.POINTER(i32)] * 5
    L.bsdfd_measured_eval.argtypes = [vp, fp, fp, i64, C.POINTER(C.c_float), fp, vp]
    L.bsdfd_measured_sample_weight.argtypes = [vp, fp, fp, fp, fp, i64, C.POINTER(C.c_float), C.c_float, fp, fp, vp]
    L.bsdfd_bucket_workspace_bytes.argtypes = [i64, i32]
    L.bsdfd_bucket_workspace_bytes.restype = i64
    L.bsdfd_bucket_by_material.argtypes = [fp, i64, i32, fp, fp, fp, i64, vp]
    L.bsdfd_gather_lanes.argtypes = [fp, i64, fp, fp, vp]
    L.bsdfd_scatter_lanes.argtypes = [fp, i64, fp, fp, fp, fp, fp, fp, vp]
    L.bsdfd_set_profiling.argtypes = [vp, i32]
    L.bsdfd_profile_read.argtypes = [vp, C.POINTER(i64), C.POINTER(C.c_double)]
    L.bsdfd_profile_read_op.argtypes = [vp, C.c_int32, C.POINTER(i64), C.POINTER(C.c_double)]
    L.bsdfd_profile_clock_mhz.argtypes = [vp, C.POINTER(C.c_double)]
    L.bsdfd_last_kernel_ms.argtypes = [vp]
    L.bsdfd_last_kernel_ms.restype = C.c_float
    L.bsdfd_shader_clock_mhz.argtypes = [C.POINTER(C.c_double), vp]
    L.bsdfd_last_error.restype = C.c_char_p
    L.bsdfd_version.restype = C.c_char_p
    for name in EXPORTS:
        getattr(L, name)  # AttributeError if a declared symbol is missing
    L.bsdfd_abi_version.restype = i32
    if L.bsdfd_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} implements ABI {L.bsdfd_abi_version()} of include/bsdfd.h, this host was written for "
                           f"{ABI_VERSION}: rebuild the library (__graft_entry__.build())")
    _lib = L
    return L


def shader_clock_mhz(stream=None) -> float:
    """Shader clock (MHz) under the flow kernel's instruction mix (bsdfd_shader_clock_mhz, csrc/clock.hip)."""
    mhz = C.c_double()
    check(lib().bsdfd_shader_clock_mhz(C.byref(mhz), stream))
    return mhz.value


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(f"bsdfd error {rc}: {lib().bsdfd_last_error().decode()}")
