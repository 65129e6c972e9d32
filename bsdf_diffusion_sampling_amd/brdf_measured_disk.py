"""Disk-domain measured-BRDF plugin — mirror of rendering/brdf_measured_disk.py:31-130.

``MyBSDF(props)`` reads ``props["filename"]`` (material name, :34), loads the rectified
flow net + the conditional Gaussian base net (:43-51; here from the neutral ``.bsdfw``
set shipped under data/weights/, or from ``props["checkpoint_dir"]`` pickles), and exposes
``sample / eval / pdf / eval_pdf / to_string`` with the reference's semantics:

  sample (:59-101)  omega_i = wi.xy; 4 Euler steps; r^2 >= 0.995 -> wo=(0,0,1), pdf=0;
                    z = sqrt(relu(1-r^2)); pdf_solid_angle = pdf_disk * cos(theta_o);
                    weight = f/pdf; pdf := 0 where lum(weight) >= 30;
                    lanes with cos(theta_i) <= 0, pdf <= 0 or cos(theta_o) <= 0 get weight 0.
  pdf    (:112-124) network_pdf_disk on the xy components, * cos(theta_o), cos masks.
  eval   (:103-110) ground-truth measured BSDF (NOT neural), cos masks.

Mitsuba's ``sample1/sample2`` are ignored exactly as in the reference (:59; randomness
comes from the in-kernel Philox stream keyed from torch's global generator).
"""
from __future__ import annotations

import torch

from . import _lib
from . import weights as W
from .plugin_base import (FLAG_DELTA_REFLECTION, FLAG_FRONT_SIDE, BSDFSample3f, NeuralBSDFCore, _wi_of, rgb2lum)


class MyBSDF(NeuralBSDFCore):
    DOMAIN = W.DOMAIN_DISK
    DOMAIN_NAME = "disk"
    VARIANT = _lib.PLUGIN_MEASURED
    T = 4
    FIREFLY = 30.0

    def __init__(self, props):
        super().__init__(props)
        self.m_flags = FLAG_DELTA_REFLECTION | FLAG_FRONT_SIDE  # :55-57
        self.m_components = [self.m_flags]

    def sample(self, ctx, si, sample1=None, sample2=None, active=True, *, x0=None, seed=None):
        wi = _wi_of(si)
        wo, pdf_sa = self.sample_t(wi, x0=x0, seed=seed)
        bs = BSDFSample3f(wo=wo, pdf=pdf_sa, eta=1.0, sampled_type=self.m_flags, sampled_component=0)
        if self.bsdf is None:  # no ground-truth evaluator: the sampler-only use (bench / harness)
            return bs, None
        if self._native_gt() is not None:  # weight, firefly rule and masks fused into the evaluator's launch
            weight, bs.pdf = self.bsdf.sample_weight(wi, wo, pdf_sa, tint=self.albedo, firefly_threshold=self.FIREFLY,
                                                     active=None if active is True else torch.as_tensor(active, device=wi.device))
            return bs, weight
        act = (wi[:, 2] > 0) if active is True else (torch.as_tensor(active, device=wi.device) & (wi[:, 2] > 0))
        value = self.eval_unmasked(ctx, si, wo) / pdf_sa[:, None]
        bs.pdf = self.apply_firefly_clamp(pdf_sa, rgb2lum(value), self.FIREFLY)
        keep = act & (bs.pdf > 0) & (wo[:, 2] > 0)
        return bs, torch.where(keep[:, None], value, torch.zeros_like(value))

    def eval_unmasked(self, ctx, si, wo):
        from .plugin_base import _vec
        return _vec(self._need_bsdf().eval(ctx, si, wo)) * self.albedo.to(wo.device)


if __name__ == "__main__":  # rendering/brdf_measured_disk.py:__main__ — render with this plugin
    from .render_cli import main
    main(MyBSDF, "diffusion_brdf_measured_disk/material_ball")
