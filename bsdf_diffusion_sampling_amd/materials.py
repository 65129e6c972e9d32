"""Mixed-material batches (BASELINE.json configs[3]: "all paper measured BSDFs, mixed queries").

The reference binds one plugin instance per material and Mitsuba dispatches each wavefront
lane to its instance (one `sample()` call per material per bounce,
rendering/matpreview/disney_bsdf_array0_envmap.xml: 12 `mybsdf` instances).  Here a
``MaterialTable`` holds one packed device handle per material and serves a batch whose
queries carry a material id: queries are bucketed (stable sort by id), each non-empty
bucket is ONE fused kernel launch on its contiguous slice, and results are scattered back
to the callers' order.  All disk nets together are ~160 KB of fp16 fragments — the whole
LDS — so keeping every material resident in one launch is not an option (SURVEY.md §7
"mixed-material batches"); per-bucket launches keep each workgroup's LDS image to one
material (16 KB) and lose nothing once buckets are >> 64 K queries.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from . import weights as W
from .sampler import FlowSampler
from .sharding import bucket_by_material


class MaterialTable:
    def __init__(self, stems: Sequence[str], precision: str = "default"):
        """``stems`` are shipped weight-set names such as ``aniso_miro_7_rgb_disk`` or
        ``chm_orange_rgb_spherical`` (mixing domains is allowed: the domain, T and plugin
        variant are per material)."""
        self.stems = list(stems)
        self.samplers: List[FlowSampler] = []
        self.T: List[int] = []
        self.variant: List[int] = []
        for stem in self.stems:
            fw = W.load(W.shipped_path(*self._split(stem)))
            self.samplers.append(FlowSampler(fw, precision=precision))
            disk = fw.domain == W.DOMAIN_DISK
            self.T.append(4 if disk else 8)  # plugin defaults, brdf_measured_disk.py:68 / _spherical.py:78
            self.variant.append(_lib.PLUGIN_FULLSPHERE if stem.startswith("bsdf_") else _lib.PLUGIN_MEASURED)

    @staticmethod
    def _split(stem: str) -> Tuple[str, str]:
        for dom in ("disk", "spherical"):
            if stem.endswith("_" + dom):
                return stem[: -len(dom) - 1], dom
        raise ValueError(f"cannot parse weight-set name {stem!r}")

    @classmethod
    def all_measured(cls, precision: str = "default") -> "MaterialTable":
        """27 disk + 25 spherical measured materials (config 4)."""
        stems = W.list_shipped("disk") + [s for s in W.list_shipped("spherical")
                                          if not s.startswith("bsdf_") and not s.endswith("_complex")]
        return cls([s for s in stems if not s.endswith("_complex")], precision)

    def __len__(self):
        return len(self.samplers)

    def _buckets(self, material_id: torch.Tensor):
        if material_id.dtype != torch.int64:
            material_id = material_id.long()
        perm, counts = bucket_by_material(material_id, len(self))
        return perm, counts.cpu().tolist()

    def sample(self, material_id: torch.Tensor, wi: torch.Tensor, seed: int = 0, offset: int = 0,
               T: Optional[int] = None, x0: Optional[torch.Tensor] = None):
        """wi [N,3], material_id [N] -> (wo [N,3], pdf_sa [N]) in the callers' order.
        The Philox counter of a query is ``offset + its position in the caller's batch``... after
        bucketing positions change, so the stream is keyed per bucket: counter = offset + rank of
        the query inside its bucket, seed mixed with the material index."""
        perm, counts = self._buckets(material_id)
        wi_s = wi[perm].contiguous()
        x0_s = None if x0 is None else x0[perm].contiguous()
        wo_s = torch.empty_like(wi_s)
        pdf_s = torch.empty(wi_s.shape[0], dtype=torch.float32, device=wi.device)
        lo = 0
        for m, n in enumerate(counts):
            if n == 0:
                continue
            sl = slice(lo, lo + n)
            self.samplers[m].plugin_sample(wi_s[sl], None if x0_s is None else x0_s[sl],
                                           T=self.T[m] if T is None else T, variant=self.variant[m],
                                           seed=seed * 1000003 + m, offset=offset, out=(wo_s[sl], pdf_s[sl]))
            lo += n
        wo = torch.empty_like(wo_s)
        pdf = torch.empty_like(pdf_s)
        wo[perm] = wo_s
        pdf[perm] = pdf_s
        return wo, pdf

    def pdf(self, material_id: torch.Tensor, wi: torch.Tensor, wo: torch.Tensor, T: Optional[int] = None):
        perm, counts = self._buckets(material_id)
        wi_s, wo_s = wi[perm].contiguous(), wo[perm].contiguous()
        pdf_s = torch.empty(wi_s.shape[0], dtype=torch.float32, device=wi.device)
        lo = 0
        for m, n in enumerate(counts):
            if n == 0:
                continue
            sl = slice(lo, lo + n)
            self.samplers[m].plugin_pdf(wi_s[sl], wo_s[sl], T=self.T[m] if T is None else T,
                                        variant=self.variant[m], out=pdf_s[sl])
            lo += n
        pdf = torch.empty_like(pdf_s)
        pdf[perm] = pdf_s
        return pdf
