"""``positional_encoding_1`` of the reference (rendering/utils/model.py:9-57) as a stand-alone
GPU pass over libbsdfd.so (csrc/encoding.hip).  Inside the flow kernel the encoding is fused;
this form is the drop-in for callers of the reference function (same signature, same column
order) and the "encoding pass" whose HBM rate bench.py reports.  No CPU fallback."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def positional_encoding_1(tensor: torch.Tensor, num_encoding_functions: int = 6, include_input: bool = True,
                          log_sampling: bool = True, out: torch.Tensor = None) -> torch.Tensor:
    if not tensor.is_cuda:
        raise RuntimeError("positional_encoding_1: the HIP path needs a CUDA(ROCm) tensor; there is no CPU fallback")
    if tensor.dtype != torch.float32:
        raise TypeError("positional_encoding_1: fp32 only (the reference sets the default dtype to fp32)")
    if num_encoding_functions == 0 and include_input:
        return tensor  # "Special case, for no positional encoding" (model.py:52-53)
    x = tensor.contiguous()
    dim = x.shape[-1]
    rows = x.numel() // dim
    row_len = dim * (int(include_input) + 2 * num_encoding_functions)
    shape = tuple(x.shape[:-1]) + (row_len,)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
        raise ValueError("positional_encoding_1: bad `out`")
    with torch.cuda.device(x.device):
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(_lib.lib().bsdfd_positional_encoding(C.c_void_p(x.data_ptr()), rows, dim, num_encoding_functions,
                                                        int(include_input), int(log_sampling),
                                                        C.c_void_p(out.data_ptr()), stream))
    return out
