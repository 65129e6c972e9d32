"""Shared machinery of the three ``MyBSDF`` plugin mirrors.

The reference's plugins are ``mi.BSDF`` subclasses whose ``sample``/``pdf`` bodies are
tensor code between two DrJit<->torch hand-offs (rendering/brdf_measured_disk.py:59-124).
Here the tensor core is Mitsuba-free (``sample_t`` / ``pdf_t`` — one fused kernel launch
each, warps and guards included) and the ``mi.BSDF`` protocol methods are thin shims over
it, so the same class serves a torch-only host (tests, bench, the wavefront harness) and,
when ``mitsuba`` is importable, the Mitsuba adapter (mitsuba_adapter.py).
"""
from __future__ import annotations

import dataclasses
import os
from typing import Any, Optional

import torch

from . import _lib
from . import weights as W
from .sampler import FlowSampler

# mi.BSDFFlags values used by the reference (rendering/brdf_measured_disk.py:55,
# rendering/bsdf_myresult.py:56,90)
FLAG_DIFFUSE_REFLECTION = 0x8
FLAG_DIFFUSE_TRANSMISSION = 0x10
FLAG_DELTA_REFLECTION = 0x100
FLAG_FRONT_SIDE = 0x10000
FLAG_BACK_SIDE = 0x20000


@dataclasses.dataclass
class BSDFSample3f:
    """Torch-side stand-in for ``mi.BSDFSample3f`` (fields the reference fills, :76-87)."""
    wo: torch.Tensor
    pdf: torch.Tensor
    eta: Any = 1.0
    sampled_type: Any = 0
    sampled_component: int = 0


@dataclasses.dataclass
class SurfaceInteraction:
    """Minimal ``si``: the plugins only read ``si.wi`` ([N,3], local shading frame)."""
    wi: torch.Tensor


def rgb2lum(rgb: torch.Tensor) -> torch.Tensor:
    """rendering/utils/mitsuba_brdf_draw.py:36-38."""
    return 0.2126 * rgb[..., 0] + 0.7152 * rgb[..., 1] + 0.0722 * rgb[..., 2]


def _wi_of(si) -> torch.Tensor:
    wi = si.wi if hasattr(si, "wi") else si
    if not isinstance(wi, torch.Tensor):
        wi = wi.torch()  # DrJit array
    return wi.detach().to(torch.float32).contiguous()


def _vec(v) -> torch.Tensor:
    if not isinstance(v, torch.Tensor):
        v = v.torch()
    return v.detach().to(torch.float32).contiguous()


class NeuralBSDFCore:
    """Weights + the fused sampler for one material; subclasses fix domain/variant/T."""

    DOMAIN = W.DOMAIN_DISK
    DOMAIN_NAME = "disk"
    VARIANT = _lib.PLUGIN_MEASURED
    T = 4

    def __init__(self, props):
        self.props = props
        get = (lambda k, d=None: props[k] if k in props else d) if hasattr(props, "__contains__") else \
              (lambda k, d=None: getattr(props, k, d))
        self._get = get
        self.material = self._material_name()
        self.precision = get("precision", "default")
        self.T = int(get("T", type(self).T))
        # ground-truth evaluator (object with .eval(ctx, si, wo) -> [N,3]): given, or the native RGL
        # evaluator when measuredbsdfs/<material>.bsdf is found (rendering/brdf_measured_disk.py:36-42)
        self.bsdf = get("bsdf", None)
        if self.bsdf is None and get("measured", True):
            from .measured import MeasuredBSDF, find_measured_file
            path = get("measured_file", None) or find_measured_file(self._gt_name(), get("measured_dir", None))
            if path is not None:
                self.bsdf = MeasuredBSDF(path)
        self.albedo = torch.tensor(get("albedo", [1.0, 1.0, 1.0]), dtype=torch.float32)
        fw = self._load_weights(get("weights", None), get("checkpoint_dir", None))
        # props["tile"]: bsdfd_desc.tile — 0 (library default: 32-query tiles for these nets), 16 or 32 queries per wave tile
        self.sampler = FlowSampler(fw, precision=self.precision, tile=int(get("tile", 0)))
        # per-query context (include/bsdfd.h, bsdfd_context_bytes): a renderer asks pdf(si, wl) and sample(si) for the same
        # intersections (rendering/brdf_measured_disk.py:112,59; Mitsuba's path integrator in the order eval_pdf() -> sample()),
        # so whichever call sees an si.wi first also writes what depends on wi alone and the later ones read it.  Keyed on the
        # identity + version of the wi tensor and on the sampler handle; 144 B per query (ONE reused buffer per plugin
        # instance).  OPT-IN since round 4 (props["context_cache"] = True): with the cheaper prologue of that round a
        # sample()+pdf() pair of a 1 Mi-query wavefront gains 1.1 % (disk, T = 8), 2.2 % (spherical) and nothing measurable at the
        # disk plugin's default T = 4 (tools/ctx_pair.py, profiles/r04_ab/ctx_pair.txt) for 144 B of device memory per query and
        # 5x the HBM traffic.  Capped at 192 MiB: above that the record streams through HBM instead of the 256 MiB Infinity
        # Cache and stops paying altogether.
        self.context_cache = bool(get("context_cache", False))
        self.context_cache_max_bytes = int(get("context_cache_max_bytes", 192 << 20))
        # (key, wi tensor, buffer) of the last launch that FILLED the buffer successfully.  The entry keeps a reference to the
        # tensor it was filled for (12 B per query next to the 144 B of the record): its storage cannot be freed and handed to
        # another tensor — same pointer, version 0 again — while the entry lives.
        self._ctx = None
        self._ctx_buf = None
        self._ctx_event = None  # recorded behind every launch that touched the buffer (cross-stream ordering)
        self._ctx_stream = None
        # a host that hands every call a fresh copy of wi (e.g. a separate `si.wi.torch()` per method) never hits: after
        # `context_cache_patience` fills in a row that nobody read, filling stops (it costs ~1 % of a launch) and is retried
        # once every 64 calls
        self.context_cache_patience = int(get("context_cache_patience", 4))
        self._ctx_unread_fills = 0
        self._ctx_skipped = 0

    # -- weight discovery ------------------------------------------------
    def _material_name(self) -> str:
        return str(self._get("filename"))

    def _gt_name(self) -> str:
        """File stem of the ground-truth tensor file (`measuredbsdfs/<stem>.bsdf`)."""
        return self.material

    def _ckpt_tag(self) -> str:
        return self.material

    def _load_weights(self, path: Optional[str], ckpt_dir: Optional[str]) -> W.FlowWeights:
        if path is None:
            path = W.shipped_path(self.material, self.DOMAIN_NAME)
        if os.path.exists(path):
            return W.load(path)
        if ckpt_dir is not None:
            # the reference's pickle layout: checkpoints_new/<mat>_<domain>/brdf_{rectify,pretrain}_network<tag>.pth
            d = os.path.join(ckpt_dir, f"{self.material}_{self.DOMAIN_NAME}")
            tag = self._ckpt_tag()
            sd = torch.load(os.path.join(d, f"brdf_rectify_network{tag}.pth"), map_location="cpu")
            bd = torch.load(os.path.join(d, f"brdf_pretrain_network{tag}.pth"), map_location="cpu")
            return W.from_state_dicts(self.material, self.DOMAIN, sd, bd)
        raise FileNotFoundError(f"no weights for material {self.material!r} ({self.DOMAIN_NAME}): {path}")

    # -- tensor core -------------------------------------------------------
    def sample_t(self, wi: torch.Tensor, x0: Optional[torch.Tensor] = None, seed: Optional[int] = None,
                 offset: int = 0):
        """wi [N,3] -> (wo [N,3], pdf_sa [N]) with the warp and guards of sample() fused;
        the firefly rule is separate (``apply_firefly_clamp``) because it needs eval()."""
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        cin, cout, key = self._ctx_lookup(wi)
        res = self.sampler.plugin_sample(wi, x0, T=self.T, variant=self.VARIANT, seed=seed, offset=offset, ctx_out=cout, ctx_in=cin)
        self._ctx_done(wi, key, cin, cout)
        return res

    def pdf_t(self, wi: torch.Tensor, wo: torch.Tensor) -> torch.Tensor:
        cin, cout, key = self._ctx_lookup(wi)
        res = self.sampler.plugin_pdf(wi, wo, T=self.T, variant=self.VARIANT, ctx_in=cin, ctx_out=cout)
        self._ctx_done(wi, key, cin, cout)
        return res

    # -- context cache -----------------------------------------------------
    def _wi_key(self, wi: torch.Tensor):
        return (wi.data_ptr(), wi._version, wi.shape[0], wi.device, id(self.sampler), getattr(self.sampler, "_hi", None))

    def _ctx_lookup(self, wi):
        """(ctx_in, ctx_out, key) for a launch on ``wi``: a hit reads the buffer the last filling launch wrote for this very
        tensor (any of sample_t / pdf_t, in any order); a miss hands out the buffer to fill — it becomes the cached context
        only once the launch has returned without raising (``_ctx_done``).  (None, None, None): cache off, not a device
        tensor, or the wavefront is over the cap."""
        if not self.context_cache or not isinstance(wi, torch.Tensor) or wi.dim() != 2 or wi.shape[0] == 0 or not wi.is_cuda:
            return None, None, None
        key = self._wi_key(wi)
        cur = torch.cuda.current_stream(wi.device)
        c = self._ctx
        if c is not None and c[0] == key:
            self._ctx_order(cur)       # the filling launch may have run on another stream
            self._ctx_unread_fills = 0
            return c[2], None, key
        if self._ctx_unread_fills >= self.context_cache_patience:
            self._ctx_skipped += 1
            if self._ctx_skipped % 64:
                self._ctx = None
                return None, None, None
        need = self.sampler.context_floats(wi.shape[0])
        if need * 4 > self.context_cache_max_bytes:
            self._ctx = None
            return None, None, None
        buf = self._ctx_buf
        if buf is None or buf.numel() < need or buf.device != wi.device:
            buf = self._ctx_buf = torch.empty((need,), dtype=torch.float32, device=wi.device)
            self._ctx_event = None
        # about to be overwritten: whatever launch — on whatever stream — still reads the previous contents goes first, and
        # the old entry is dropped NOW (if this launch raises, the buffer holds neither the old nor the new context)
        self._ctx = None
        self._ctx_order(cur)
        return None, buf, key

    def _ctx_order(self, cur):
        if self._ctx_event is not None and self._ctx_stream != cur:
            cur.wait_event(self._ctx_event)

    def _ctx_done(self, wi, key, cin, cout):
        """Behind a launch that used the buffer: commit a fill (the key also pins the tensor object: a freed-and-reallocated
        wi with the same address cannot alias it) and record the event later launches on other streams order against."""
        if cin is None and cout is None:
            return
        if cout is not None:
            self._ctx = (key, wi, cout)
            self._ctx_unread_fills += 1
        cur = torch.cuda.current_stream(wi.device)
        if self._ctx_event is None:
            self._ctx_event = torch.cuda.Event()
        self._ctx_event.record(cur)
        self._ctx_stream = cur

    def invalidate_context(self):
        """Drop the cached context.  The cache trusts torch's version counter: code that rewrites the ``wi`` storage
        through a raw pointer (a native kernel, DLPack) must either bump it
        (``torch.autograd.graph.increment_version(wi)``), call this, or construct the plugin with
        ``context_cache=False``."""
        self._ctx = None

    def sample_pdf_t(self, wi: torch.Tensor, wl: torch.Tensor, x0: Optional[torch.Tensor] = None,
                     seed: Optional[int] = None, offset: int = 0):
        """``sample_t(wi)`` and ``pdf_t(wi, wl)`` of the same intersections in one launch (a renderer with
        next-event estimation asks both per path): -> (wo [N,3], pdf(wo) [N], pdf(wl) [N])."""
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        return self.sampler.plugin_sample_pdf(wi, wl, x0, T=self.T, variant=self.VARIANT, seed=seed, offset=offset)

    @staticmethod
    def apply_firefly_clamp(pdf: torch.Tensor, weight_lum: torch.Tensor, thr: float) -> torch.Tensor:
        """``pdf = where(lum(f/pdf) < thr, pdf, 0)`` (rendering/brdf_measured_disk.py:97-100 thr=30,
        brdf_measured_spherical.py:106-108 thr=30, bsdf_myresult.py:100-103 thr=3.5)."""
        return torch.where(weight_lum < thr, pdf, torch.zeros_like(pdf))

    # -- mi.BSDF protocol ----------------------------------------------------
    def _need_bsdf(self):
        if self.bsdf is None:
            raise RuntimeError("eval() delegates to a ground-truth BSDF (Mitsuba's `measured` plugin in the "
                               "reference, rendering/brdf_measured_disk.py:36-42); pass props['bsdf']")
        return self.bsdf

    def _native_gt(self):
        from .measured import MeasuredBSDF
        return self.bsdf if isinstance(self.bsdf, MeasuredBSDF) else None

    def eval(self, ctx, si, wo, active=True):
        wi, wo_t = _wi_of(si), _vec(wo)
        if self._native_gt() is not None:  # one launch: f cos * albedo, zero on the lower hemispheres
            return self.bsdf.eval_t(wi, wo_t, tint=self.albedo)
        value = _vec(self._need_bsdf().eval(ctx, si, wo)) * self.albedo.to(wi.device)
        ok = (wi[:, 2] > 0) & (wo_t[:, 2] > 0)
        return torch.where(ok[:, None], value, torch.zeros_like(value))

    def pdf(self, ctx, si, wo, active=True):
        return self.pdf_t(_wi_of(si), _vec(wo))

    def eval_pdf(self, ctx, si, wo, active=True):
        return self.eval(ctx, si, wo, active), self.pdf(ctx, si, wo, active)

    def to_string(self):
        return "MyBSDF[\n" "    albedo=%s,\n" "]" % (self.albedo.tolist(),)
