"""Operator surface of the hot path — same four free functions, names, argument
order and defaults as the reference's rendering/utils/mlp_brdf_sampling.py:

    network_sampling_disk(D_base, D_sample, omega_i, T=4)            -> (x [N,2], pdf [N])   :17-51
    network_pdf_disk(D_base, D_sample, omega_o, omega_i, T=4)        -> pdf [N]              :69-103
    network_sampling_spherical(D_base, D_sample, omega_i, T=8)       -> (x [N,2], pdf [N])   :106-140
    network_pdf_spherical(D_base, D_sample, omega_o, omega_i, T=8)   -> pdf [N]              :144-181

``D_base`` / ``D_sample`` are the reference-shaped containers of ``model.py`` (or a
ready ``FlowSampler`` passed as ``D_sample`` with ``D_base=None``).  The pair is packed
into MFMA fragments once and the device handle is cached on ``D_sample``; each call is
then ONE fused kernel launch on the current stream instead of the reference's
~100 eager launches + 2 autograd backward passes per Euler step.

Differences from the reference, all deliberate:
  * D_base is evaluated once per query (the reference evaluates it twice, :20,:24);
  * extra keyword-only arguments: ``x0`` injects the base draw (parity tests), ``seed``
    / ``offset`` key the in-kernel Philox stream; by default the seed is drawn from
    torch's global generator, so ``torch.manual_seed`` controls reproducibility as it
    does for the reference's ``torch.randn_like``;
  * outputs carry no autograd graph (the reference's pdf keeps a ``grad_fn``).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import weights as W
from .sampler import FlowSampler


def _packed(D_base, D_sample, domain: int, precision: str) -> FlowSampler:
    if isinstance(D_sample, FlowSampler):
        if D_sample.domain != domain:
            raise RuntimeError("FlowSampler domain does not match the operator called")
        return D_sample
    from . import model as M
    # The packed device handle is cached on D_sample, keyed by the IDENTITY AND VERSION of every weight tensor of
    # the pair: (data_ptr, torch's in-place version counter) per state-dict entry.  An optimiser step, a
    # load_state_dict or any other in-place update of a real reference nn.Module bumps `_version` (and a rebind
    # changes data_ptr), so a training loop that samples from a net it is updating (the reflow stage) never sees
    # stale packed weights; a frozen net costs one tuple of ~10 integers per call.
    def stamp(mod):
        # named_parameters / named_buffers: no state_dict() is built per call (this path is host-bound at ~8 us)
        return (id(mod),) + tuple((k, v.data_ptr(), v._version) for k, v in
                                  list(mod.named_parameters(recurse=True)) + list(mod.named_buffers(recurse=True)))
    # id(module) is part of the key: a REPLACED module or parameter whose storage reuses a freed address with the same
    # version counter must not hit a stale entry (the cache lives on D_sample, so id(D_sample) is implied; D_base is not)
    key = (stamp(D_base), stamp(D_sample), domain, precision, torch.cuda.current_device())
    cache = D_sample.__dict__.setdefault("_bsdfd_cache", {})
    s = cache.get(key)
    if s is None:
        cache.clear()
        s = FlowSampler(M.to_flow_weights(D_base, D_sample, domain), precision=precision)
        cache[key] = s
    return s


def repack(D_sample) -> None:
    """Drop the cached device handle of ``D_sample`` (next call re-packs).  Only needed after writes torch's
    version counter cannot see, i.e. through ``param.data`` (``p.data.mul_(...)``); optimiser steps,
    ``load_state_dict`` and ``with torch.no_grad(): p.copy_/add_(...)`` are detected automatically."""
    D_sample.__dict__.pop("_bsdfd_cache", None)


def _seed(seed: Optional[int]) -> int:
    if seed is not None:
        return int(seed)
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def _f32(t: torch.Tensor) -> torch.Tensor:
    # the reference casts omega_o explicitly (:71,:146); do the same for all inputs
    return t.detach().to(dtype=torch.float32).contiguous()


def network_sampling_disk(D_base, D_sample, omega_i, T=4, *, x0=None, seed=None, offset=0, precision="default"):
    s = _packed(D_base, D_sample, W.DOMAIN_DISK, precision)
    return s.network_sampling(_f32(omega_i), None if x0 is None else _f32(x0), T=T, seed=_seed(seed), offset=offset)


def network_pdf_disk(D_base, D_sample, omega_o, omega_i, T=4, *, precision="default"):
    s = _packed(D_base, D_sample, W.DOMAIN_DISK, precision)
    return s.network_pdf(_f32(omega_o), _f32(omega_i), T=T)


def network_sampling_spherical(D_base, D_sample, omega_i, T=8, *, x0=None, seed=None, offset=0,
                               precision="default"):
    s = _packed(D_base, D_sample, W.DOMAIN_SPHERICAL, precision)
    return s.network_sampling(_f32(omega_i), None if x0 is None else _f32(x0), T=T, seed=_seed(seed), offset=offset)


def network_pdf_spherical(D_base, D_sample, omega_o, omega_i, T=8, *, precision="default"):
    s = _packed(D_base, D_sample, W.DOMAIN_SPHERICAL, precision)
    return s.network_pdf(_f32(omega_o), _f32(omega_i), T=T)


def flow_samples_only(D_base, D_sample, domain: int, x0, omega_i, T, *, precision="f16"):
    """Reflow teacher sampling (no Jacobian): the reference's only tiny-cuda-nn call site,
    ``dosampling`` in learning_repo_cleanup/spherical_domain_sampling.py:147-166 and
    disk_domain_sampling.py:93-110 (T = 128 / 256 Euler steps, fp16 FullyFusedMLP)."""
    s = _packed(D_base, D_sample, domain, precision)
    return s.flow_samples_only(_f32(omega_i), _f32(x0), T=T)
