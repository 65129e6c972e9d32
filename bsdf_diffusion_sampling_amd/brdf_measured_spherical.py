"""Spherical-domain measured-BRDF plugin — mirror of rendering/brdf_measured_spherical.py:40-143.

State is (theta, phi); T = 8 Euler steps; the net sees [theta, sin phi, cos phi].

  sample (:69-109)  (theta_i, phi_i) = cart_to_spher(wi) (:35-39, acos(z/(r+1e-8)), atan2);
                    pdf := 0 where sin(theta_o) <= 5e-5 or cos(theta_o) <= 0;
                    wo = sph_to_dir(theta, phi) (:31-34);
                    pdf_solid_angle = pdf * clamp(1/sin_theta(wo), 1, FLT_MAX) (:89-91);
                    firefly rule with threshold 30 (:106-108).
  pdf    (:122-137) cart_to_spher on both; sin guard; * clamp(1/sin theta_o, 1, FLT_MAX); cos masks.

Reference bug not reproduced: the reference loads the ``_disk`` pretrain
checkpoint into the spherical base net (:59); this mirror pairs ``_spherical`` with
``_spherical``.  Pass ``props["weights"]`` to override.
"""
from __future__ import annotations

import torch

from . import _lib
from . import weights as W
from .plugin_base import (FLAG_DELTA_REFLECTION, FLAG_FRONT_SIDE, BSDFSample3f, NeuralBSDFCore, _vec, _wi_of,
                          rgb2lum)


def cart_to_spher(xyz: torch.Tensor) -> torch.Tensor:
    """Same map the kernels fuse; exposed for callers that want (theta, phi) themselves."""
    r = torch.linalg.vector_norm(xyz, dim=1)
    return torch.stack([torch.acos(xyz[:, 2] / (r + 1e-8)), torch.atan2(xyz[:, 1], xyz[:, 0])], dim=1)


class MyBSDF(NeuralBSDFCore):
    DOMAIN = W.DOMAIN_SPHERICAL
    DOMAIN_NAME = "spherical"
    VARIANT = _lib.PLUGIN_MEASURED
    T = 8
    FIREFLY = 30.0

    def __init__(self, props):
        super().__init__(props)
        self.m_flags = FLAG_DELTA_REFLECTION | FLAG_FRONT_SIDE  # :63-65
        self.m_components = [self.m_flags]

    def sample(self, ctx, si, sample1=None, sample2=None, active=True, *, x0=None, seed=None):
        wi = _wi_of(si)
        wo, pdf_sa = self.sample_t(wi, x0=x0, seed=seed)
        bs = BSDFSample3f(wo=wo, pdf=pdf_sa, eta=1.0, sampled_type=self.m_flags, sampled_component=0)
        if self.bsdf is None:
            return bs, None
        if self._native_gt() is not None:  # weight, firefly rule and masks fused into the evaluator's launch
            weight, bs.pdf = self.bsdf.sample_weight(wi, wo, pdf_sa, tint=self.albedo, firefly_threshold=self.FIREFLY,
                                                     active=None if active is True else torch.as_tensor(active, device=wi.device))
            return bs, weight
        act = (wi[:, 2] > 0) if active is True else (torch.as_tensor(active, device=wi.device) & (wi[:, 2] > 0))
        value = _vec(self._need_bsdf().eval(ctx, si, wo)) * self.albedo.to(wo.device) / pdf_sa[:, None]
        value = torch.where((act & (pdf_sa > 0))[:, None], value, torch.zeros_like(value))  # :105
        bs.pdf = self.apply_firefly_clamp(pdf_sa, rgb2lum(value), self.FIREFLY)
        keep = act & (bs.pdf > 0) & (wo[:, 2] > 0)
        return bs, torch.where(keep[:, None], value, torch.zeros_like(value))


if __name__ == "__main__":  # rendering/brdf_measured_spherical.py:__main__ — render with this plugin
    from .render_cli import main
    main(MyBSDF, "diffusion_brdf_measured_spherical/material_ball")
