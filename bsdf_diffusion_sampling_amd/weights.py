"""Neutral packed weight file (``.bsdfw``) for one (material, domain) flow sampler.

Replaces the reference's pickle ``.pth`` state-dicts
(reference: rendering/brdf_measured_disk.py:43-51,
rendering/brdf_measured_spherical.py:53-59, rendering/bsdf_myresult.py:49-54,
saved by learning_repo_cleanup/utils/utils.py:30-32).  A ``.bsdfw`` file is
pure data: a fixed little-endian header followed by fp32 blobs, readable from
C (``bsdfd_create_from_file`` in include/bsdfd.h) and from numpy without torch.

Layout (little endian)::

    0   8 B   magic  b"BSDFWT01"
    8   64 B  material name, zero padded utf-8
    72  8 x i32: domain (0 disk, 1 spherical), width, n_hidden, pe_bands,
                 base_hidden, base_pe_bands, state_dim (2 disk / 3 spherical),
                 reserved(0)
    104 f32   w_in     [width, in_dim]     in_dim = state_dim + 1 + 2 + 4*pe_bands
        f32   w_hidden [n_hidden-1, width, width]
        f32   w_out    [2, width]
        f32   base_w1  [base_hidden, 2 + 4*base_pe_bands]
        f32   base_b1  [base_hidden]
        f32   base_w2  [4, base_hidden]
        f32   base_b2  [4]

All matrices are row-major ``[out, in]`` exactly as ``nn.Linear.weight`` stores
them.  Column order of ``w_in`` follows the reference's
concatenation ``[state | alpha | PE(omega_i)]`` (rendering/utils/model.py:494-495).
"""
from __future__ import annotations

import dataclasses
import os
import struct
from typing import Dict, Optional

import numpy as np

MAGIC = b"BSDFWT01"
DOMAIN_DISK = 0
DOMAIN_SPHERICAL = 1
_NAME_BYTES = 64
_HEADER = struct.Struct("<8s64s8i")


@dataclasses.dataclass
class FlowWeights:
    """fp32 weights of one velocity net + its conditional base-density net."""

    name: str
    domain: int  # DOMAIN_DISK / DOMAIN_SPHERICAL
    width: int
    n_hidden: int
    pe_bands: int
    base_hidden: int
    base_pe_bands: int
    w_in: np.ndarray  # [width, in_dim]
    w_hidden: np.ndarray  # [n_hidden-1, width, width]
    w_out: np.ndarray  # [2, width]
    base_w1: np.ndarray  # [base_hidden, base_in]
    base_b1: np.ndarray  # [base_hidden]
    base_w2: np.ndarray  # [4, base_hidden]
    base_b2: np.ndarray  # [4]

    @property
    def state_dim(self) -> int:
        return 2 if self.domain == DOMAIN_DISK else 3

    @property
    def in_dim(self) -> int:
        return self.state_dim + 1 + 2 + 4 * self.pe_bands

    @property
    def base_in(self) -> int:
        return 2 + 4 * self.base_pe_bands

    def validate(self) -> "FlowWeights":
        w, nh = self.width, self.n_hidden
        exp = {
            "w_in": (w, self.in_dim),
            "w_hidden": (nh - 1, w, w),
            "w_out": (2, w),
            "base_w1": (self.base_hidden, self.base_in),
            "base_b1": (self.base_hidden,),
            "base_w2": (4, self.base_hidden),
            "base_b2": (4,),
        }
        if self.domain not in (DOMAIN_DISK, DOMAIN_SPHERICAL):
            raise ValueError(f"bad domain {self.domain}")
        if nh < 1:
            raise ValueError("n_hidden must be >= 1")
        for k, shp in exp.items():
            a = np.ascontiguousarray(getattr(self, k), dtype=np.float32).reshape(shp) \
                if getattr(self, k).size == int(np.prod(shp)) else None
            if a is None:
                raise ValueError(f"{k}: expected shape {shp}, got {getattr(self, k).shape}")
            setattr(self, k, a)
        return self

    def flops_per_step(self) -> int:
        """Algorithmic flop per Euler step: 2*MAC of the
        unpadded velocity forward plus the 2-tangent forward-mode Jacobian."""
        w, nh, sd = self.width, self.n_hidden, self.state_dim
        fwd = self.in_dim * w + (nh - 1) * w * w + 2 * w
        tang = 2 * ((nh - 1) * w * w + 2 * w)
        if sd == 3:  # d/dphi tangent mixes the sin/cos columns of layer 1
            tang += 2 * w
        return 2 * (fwd + tang)

    def flops_base(self) -> int:
        return 2 * (self.base_in * self.base_hidden + 4 * self.base_hidden)

    def flops_per_query(self, T: int) -> int:
        return T * self.flops_per_step() + self.flops_base()


def _blob_order():
    return ("w_in", "w_hidden", "w_out", "base_w1", "base_b1", "base_w2", "base_b2")


def save(path: str, fw: FlowWeights) -> None:
    fw.validate()
    name = fw.name.encode("utf-8")[:_NAME_BYTES]
    hdr = _HEADER.pack(MAGIC, name, fw.domain, fw.width, fw.n_hidden, fw.pe_bands,
                       fw.base_hidden, fw.base_pe_bands, fw.state_dim, 0)
    with open(path, "wb") as f:
        f.write(hdr)
        for k in _blob_order():
            f.write(np.ascontiguousarray(getattr(fw, k), dtype="<f4").tobytes())


def load(path: str) -> FlowWeights:
    with open(path, "rb") as f:
        raw = f.read()
    if len(raw) < _HEADER.size:
        raise ValueError(f"{path}: truncated header")
    magic, name, domain, width, n_hidden, pe, bh, bpe, sd, _ = _HEADER.unpack_from(raw, 0)
    if magic != MAGIC:
        raise ValueError(f"{path}: bad magic {magic!r}")
    if sd != (2 if domain == DOMAIN_DISK else 3):
        raise ValueError(f"{path}: state_dim {sd} inconsistent with domain {domain}")
    in_dim = sd + 1 + 2 + 4 * pe
    base_in = 2 + 4 * bpe
    shapes = [(width, in_dim), (n_hidden - 1, width, width), (2, width),
              (bh, base_in), (bh,), (4, bh), (4,)]
    off = _HEADER.size
    arrs = []
    for shp in shapes:
        n = int(np.prod(shp))
        if off + 4 * n > len(raw):
            raise ValueError(f"{path}: truncated payload")
        arrs.append(np.frombuffer(raw, dtype="<f4", count=n, offset=off).reshape(shp).copy())
        off += 4 * n
    if off != len(raw):
        raise ValueError(f"{path}: {len(raw) - off} trailing bytes")
    return FlowWeights(name.rstrip(b"\0").decode("utf-8"), domain, width, n_hidden, pe, bh, bpe,
                       *arrs).validate()


def from_state_dicts(name: str, domain: int, sample_sd: Dict[str, "np.ndarray"],
                     base_sd: Dict[str, "np.ndarray"], pe_bands: int = 5,
                     base_pe_bands: int = 3) -> FlowWeights:
    """Build from reference-shaped state dicts (keys ``linear{k}.weight``,
    ``output.weight`` / base ``linear1.{weight,bias}``, ``output.{weight,bias}``;
    SURVEY.md §8 a15 / Appendix C).  Values may be torch tensors or arrays."""

    def arr(v):
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        return np.asarray(v, dtype=np.float32)

    n_hidden = 0
    while f"linear{n_hidden + 1}.weight" in sample_sd:
        n_hidden += 1
    if n_hidden == 0:
        raise ValueError("state dict has no linear1.weight")
    w_in = arr(sample_sd["linear1.weight"])
    width = w_in.shape[0]
    hid = [arr(sample_sd[f"linear{k}.weight"]) for k in range(2, n_hidden + 1)]
    w_hidden = np.stack(hid) if hid else np.zeros((0, width, width), np.float32)
    base_w1 = arr(base_sd["linear1.weight"])
    return FlowWeights(name, domain, width, n_hidden, pe_bands, base_w1.shape[0], base_pe_bands,
                       w_in, w_hidden, arr(sample_sd["output.weight"]), base_w1,
                       arr(base_sd["linear1.bias"]), arr(base_sd["output.weight"]),
                       arr(base_sd["output.bias"])).validate()


# ---------------------------------------------------------------------------
# Shipped weight sets (converted once from the reference's checkpoints_new/ by
# tools/export_weights.py; weights are data).
# ---------------------------------------------------------------------------
DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "weights")


def shipped_path(material: str, domain: str, variant: Optional[str] = None) -> str:
    """``material`` is e.g. ``aniso_miro_7_rgb`` or ``bsdf_3``; domain 'disk'|'spherical'."""
    stem = f"{material}_{domain}" + (f"_{variant}" if variant else "")
    return os.path.join(DATA_DIR, stem + ".bsdfw")


def list_shipped(domain: Optional[str] = None):
    out = []
    for fn in sorted(os.listdir(DATA_DIR)) if os.path.isdir(DATA_DIR) else []:
        if fn.endswith(".bsdfw"):
            stem = fn[:-6]
            if domain is None or stem.endswith("_" + domain):
                out.append(stem)
    return out
