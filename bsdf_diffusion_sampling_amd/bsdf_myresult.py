"""Full-sphere analytic-BSDF plugin (transmission) — mirror of rendering/bsdf_myresult.py:41-139.

Same spherical operators, different post-processing:
  ctor   (:43-57)   props["idx"] selects ``bsdf_<idx>_spherical`` weights, props["albedo"];
                    flags Diffuse | FrontSide | BackSide.
  sample (:59-104)  only the sin(theta_o) > 5e-5 guard (no cos guard: theta in [0, pi]);
                    pdf_sa = pdf * clamp(1/|sin theta_o|, 1, FLT_MAX);
                    eta = 1 above / 1.788 below the surface, sampled_type 8 / 16 (:89-90);
                    weight = f * albedo / pdf * sin(theta_o) (:97); pdf := 0 where weight.x >= 3.5 (:100-103).
  pdf    (:115-133) no guards, no cos masks: pdf * clamp(1/|sin theta_o|, 1, FLT_MAX).
  eval   (:106-113) ground truth from the analytic BSDF list, no cos masks.
"""
from __future__ import annotations

import torch

from . import _lib
from . import weights as W
from .plugin_base import (FLAG_BACK_SIDE, FLAG_DIFFUSE_REFLECTION, FLAG_DIFFUSE_TRANSMISSION, FLAG_FRONT_SIDE,
                          BSDFSample3f, NeuralBSDFCore, _vec, _wi_of)


class MyBSDF(NeuralBSDFCore):
    DOMAIN = W.DOMAIN_SPHERICAL
    DOMAIN_NAME = "spherical"
    VARIANT = _lib.PLUGIN_FULLSPHERE
    T = 8
    FIREFLY = 3.5

    def __init__(self, props):
        super().__init__(props)
        self.idx = int(self._get("idx"))
        self.m_flags = FLAG_DIFFUSE_REFLECTION | FLAG_DIFFUSE_TRANSMISSION | FLAG_FRONT_SIDE | FLAG_BACK_SIDE
        self.m_components = [self.m_flags]

    def _material_name(self) -> str:
        return f"bsdf_{int(self._get('idx'))}"

    def _ckpt_tag(self) -> str:
        return str(int(self._get("idx")))

    def sample(self, ctx, si, sample1=None, sample2=None, active=True, *, x0=None, seed=None):
        wi = _wi_of(si)
        wo, pdf_sa = self.sample_t(wi, x0=x0, seed=seed)
        up = wo[:, 2] > 0
        bs = BSDFSample3f(wo=wo, pdf=pdf_sa, eta=torch.where(up, 1.0, 1.788),
                          sampled_type=torch.where(up, FLAG_DIFFUSE_REFLECTION, FLAG_DIFFUSE_TRANSMISSION),
                          sampled_component=2)
        if self.bsdf is None:
            return bs, None
        # f * albedo / pdf * sin(theta_o) with pdf_sa = pdf / sin(theta_o) wherever the clamp is inactive
        value = _vec(self._need_bsdf().eval(ctx, si, wo)) * self.albedo.to(wo.device) / pdf_sa[:, None]
        value = torch.where((pdf_sa > 0)[:, None], value, torch.zeros_like(value))
        bs.pdf = self.apply_firefly_clamp(pdf_sa, value[:, 0], self.FIREFLY)
        return bs, torch.where((bs.pdf > 0)[:, None], value, torch.zeros_like(value))

    def eval(self, ctx, si, wo, active=True):
        return _vec(self._need_bsdf().eval(ctx, si, wo)) * self.albedo.to(_wi_of(si).device)
