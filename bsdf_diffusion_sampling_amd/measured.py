"""Ground-truth evaluator for the plugins' ``eval()``: the RGL measured-BSDF model on the GPU.

The reference builds Mitsuba's ``measured`` BSDF (``mi.load_dict({'type': 'measured', 'filename':
'measuredbsdfs/<name>.bsdf'})``, rendering/brdf_measured_disk.py:36-42) and calls its ``eval`` for the
sample weight and the firefly rule.  Mitsuba has no AMD GPU variant; ``MeasuredBSDF`` is the same
model (Dupuy & Jakob 2018) over ``libbsdfd.so`` (csrc/measured.hip) with the call shape the plugins
use: ``eval(ctx, si, wo) -> [N,3]`` = f * cos(theta_o), zero on the lower hemispheres.  Only the
``*_rgb.bsdf`` flavour is supported (the one the reference's scenes name).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib


def find_measured_file(material: str, directory: Optional[str] = None) -> Optional[str]:
    """``<dir>/<material>.bsdf`` in: the argument, $BSDFD_MEASURED_DIR, ./measuredbsdfs (the reference's
    CWD-relative convention, brdf_measured_disk.py:39) — first hit, else None."""
    for d in (directory, os.environ.get("BSDFD_MEASURED_DIR"), "measuredbsdfs"):
        if d:
            p = os.path.join(d, material + ".bsdf")
            if os.path.exists(p):
                return p
    return None


class MeasuredBSDF:
    def __init__(self, path: str):
        self.path = path
        self._h = C.c_void_p()
        _lib.check(_lib.lib().bsdfd_measured_create_from_file(path.encode(), C.byref(self._h)))
        info = [C.c_int32() for _ in range(5)]
        _lib.check(_lib.lib().bsdfd_measured_get_info(self._h, *[C.byref(i) for i in info]))
        self.n_phi, self.n_theta, iso, jac, self.reduction = (i.value for i in info)
        self.isotropic, self.jacobian = bool(iso), bool(jac)

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                _lib.lib().bsdfd_measured_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def eval_t(self, wi: torch.Tensor, wo: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        for name, t in (("wi", wi), ("wo", wo)):
            if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] == 3 and t.is_contiguous()):
                raise ValueError(f"MeasuredBSDF.eval_t: {name} must be a contiguous fp32 CUDA tensor [N,3]")
        if wi.shape != wo.shape:
            raise ValueError("MeasuredBSDF.eval_t: wi and wo differ in shape")
        if out is None:
            out = torch.empty_like(wi)
        with torch.cuda.device(wi.device):
            stream = C.c_void_p(torch.cuda.current_stream(wi.device).cuda_stream)
            _lib.check(_lib.lib().bsdfd_measured_eval(self._h, C.c_void_p(wi.data_ptr()), C.c_void_p(wo.data_ptr()),
                                                      wi.shape[0], C.c_void_p(out.data_ptr()), stream))
        return out

    # the call shape of ``mi.BSDF.eval`` as the reference's plugins use it (brdf_measured_disk.py:96,107)
    def eval(self, ctx, si, wo, active=True):
        from .plugin_base import _vec, _wi_of
        return self.eval_t(_wi_of(si), _vec(wo))
