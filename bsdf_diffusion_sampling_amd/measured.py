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

    @staticmethod
    def _check(**tensors):
        ref = None
        for name, (t, cols) in tensors.items():
            shape_ok = (t.dim() == 2 and t.shape[1] == cols) if cols else t.dim() == 1
            if not (t.is_cuda and t.dtype == torch.float32 and shape_ok and t.is_contiguous()):
                raise ValueError(f"MeasuredBSDF: {name} must be a contiguous fp32 CUDA tensor "
                                 f"[N{',' + str(cols) if cols else ''}]")
            if ref is not None and t.shape[0] != ref:
                raise ValueError(f"MeasuredBSDF: {name} has {t.shape[0]} rows, expected {ref}")
            ref = t.shape[0]

    @staticmethod
    def _tint(tint):
        if tint is None:
            return None
        vals = [float(v) for v in (tint.tolist() if hasattr(tint, "tolist") else tint)]
        return (C.c_float * 3)(*vals)

    def eval_t(self, wi: torch.Tensor, wo: torch.Tensor, out: Optional[torch.Tensor] = None, tint=None) -> torch.Tensor:
        """f(wi, wo) cos(theta_o) [* tint] -> [N,3]; zero on the lower hemispheres."""
        self._check(wi=(wi, 3), wo=(wo, 3))
        if out is None:
            out = torch.empty_like(wi)
        with torch.cuda.device(wi.device):
            stream = C.c_void_p(torch.cuda.current_stream(wi.device).cuda_stream)
            _lib.check(_lib.lib().bsdfd_measured_eval(self._h, C.c_void_p(wi.data_ptr()), C.c_void_p(wo.data_ptr()),
                                                      wi.shape[0], self._tint(tint), C.c_void_p(out.data_ptr()), stream))
        return out

    def sample_weight(self, wi: torch.Tensor, wo: torch.Tensor, pdf_sa: torch.Tensor, tint=None,
                      firefly_threshold: float = 30.0, active: Optional[torch.Tensor] = None):
        """The tail of the plugins' sample() in one launch (brdf_measured_disk.py:89-101): -> (weight [N,3],
        pdf [N]) with value = f * tint / pdf_sa, the firefly rule pdf := 0 where lum(value) >= threshold,
        and weight = 0 on lanes that are inactive, have pdf 0 or leave through the lower hemisphere."""
        self._check(wi=(wi, 3), wo=(wo, 3), pdf_sa=(pdf_sa, 0))
        act = None
        if active is not None:
            act = active.to(device=wi.device, dtype=torch.uint8).contiguous()
            if act.shape != (wi.shape[0],):
                raise ValueError("MeasuredBSDF.sample_weight: active must be [N]")
        weight, pdf = torch.empty_like(wi), torch.empty_like(pdf_sa)
        with torch.cuda.device(wi.device):
            stream = C.c_void_p(torch.cuda.current_stream(wi.device).cuda_stream)
            _lib.check(_lib.lib().bsdfd_measured_sample_weight(
                self._h, C.c_void_p(wi.data_ptr()), C.c_void_p(wo.data_ptr()), C.c_void_p(pdf_sa.data_ptr()),
                None if act is None else C.c_void_p(act.data_ptr()), wi.shape[0], self._tint(tint),
                float(firefly_threshold), C.c_void_p(weight.data_ptr()), C.c_void_p(pdf.data_ptr()), stream))
        return weight, pdf

    # the call shape of ``mi.BSDF.eval`` as the reference's plugins use it (brdf_measured_disk.py:96,107)
    def eval(self, ctx, si, wo, active=True):
        from .plugin_base import _vec, _wi_of
        return self.eval_t(_wi_of(si), _vec(wo))
