"""PyTorch-ROCm operator library ``torch.ops.bsdfd.*`` (csrc/torch_ops.cpp) — the "thin PyTorch-ROCm C++ / C-ABI
extension" of the north star, modelled on how the reference binds tiny-cuda-nn
(tiny-cuda-nn/bindings/torch/tinycudann/bindings.cpp:79-110).

``libbsdfd_torch.so`` is a dispatcher-registered operator library over the SAME C ABI the ctypes shim uses
(include/bsdfd.h); it holds no kernels.  Built in-tree by ``__graft_entry__.build()`` with g++ against torch's
headers (a plain C++ translation unit: no hipify pass, no CUDA spellings); ``load()`` registers the operators.
"""
from __future__ import annotations

import os
import subprocess

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
EXT_PATH = os.path.join(_HERE, "libbsdfd_torch.so")
SRC_PATH = os.path.join(_HERE, "csrc", "torch_ops.cpp")
OPS = ("create_from_file", "create", "destroy", "flops_per_query", "network_sampling", "network_pdf", "flow_samples_only",
       "plugin_sample", "plugin_pdf", "plugin_sample_pdf", "plugin_sample_out", "plugin_pdf_out", "context_floats",
       "plugin_sample_ex_out", "plugin_pdf_ex_out")


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile libbsdfd_torch.so (needs libbsdfd.so next to it: link-time dependency, found at run time through
    the $ORIGIN rpath)."""
    import torch
    from torch.utils import cpp_extension as ce
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build(verbose=verbose)
    hdr = os.path.join(_lib.INCLUDE_DIR, "bsdfd.h")
    if (not force and os.path.exists(EXT_PATH)
            and os.path.getmtime(EXT_PATH) >= max(os.path.getmtime(SRC_PATH), os.path.getmtime(hdr))):
        return EXT_PATH
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    tmp = f"{EXT_PATH}.tmp.{os.getpid()}"
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cmd += [f"-I{p}" for p in ce.include_paths()] + [f"-I{rocm}/include", f"-I{_lib.INCLUDE_DIR}", SRC_PATH, "-o", tmp,
                                                     f"-L{_HERE}", "-l:libbsdfd.so", f"-L{torch_lib}", "-ltorch", "-ltorch_cpu",
                                                     "-lc10", "-lc10_hip", "-ltorch_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, EXT_PATH)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return EXT_PATH


_ops = None


def available() -> bool:
    """The operator library exists and resolves (through its $ORIGIN rpath) to the C-ABI library in use — an A/B
    build selected with $BSDFD_LIB_PATH is served by the ctypes shim instead."""
    return os.path.exists(EXT_PATH) and os.path.abspath(_lib.LIB_PATH) == os.path.join(_HERE, "libbsdfd.so")


def load():
    """Register the operators (raises if the library has not been built — no fallback) and return ``torch.ops.bsdfd``."""
    global _ops
    if _ops is not None:
        return _ops
    if not os.path.exists(EXT_PATH):
        raise RuntimeError(f"{EXT_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'`")
    import torch
    _lib.lib()  # the C-ABI library first, from the same path the extension's rpath resolves to
    torch.ops.load_library(EXT_PATH)
    ns = torch.ops.bsdfd
    for name in OPS:
        getattr(ns, name)  # AttributeError if an operator failed to register
    _ops = ns
    return ns
