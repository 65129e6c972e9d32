"""Weight containers with the reference's class names, constructor signatures and
``state_dict`` keys, so the reference's checkpoints load unchanged
(``torch.load`` + ``load_state_dict``, rendering/brdf_measured_disk.py:43-51).

Reference classes mirrored (rendering/utils/model.py):
  NN_cond_pos_simpler              :479-501  (2nd definition wins) 3 hidden
  NN_cond_pos                      :422-446  4 hidden
  NN_cond_pos_spherical_complicate :449-477  6 hidden
  NN_cond_pretrain_disk_one        :374-398
  NN_cond_pretrain_spherical_one   :277-317

These are *containers*: the arithmetic of the hot path lives in the fused HIP
kernels (csrc/bsdfd.hip) and is reached through
``bsdf_diffusion_sampling_amd.mlp_brdf_sampling.network_*`` which pack a
(D_base, D_sample) pair once and cache the device handle on the modules.  Calling a
container directly is not a supported path (there is deliberately no eager fallback).
"""
from __future__ import annotations

import torch

from . import weights as W


class _Container(torch.nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - by design
        raise NotImplementedError(
            f"{type(self).__name__} is a weight container; evaluate it through "
            "bsdf_diffusion_sampling_amd.mlp_brdf_sampling.network_sampling_* / network_pdf_* "
            "(fused MI355X kernels). There is no eager fallback.")

    def _bump(self):
        self._bsdfd_version = getattr(self, "_bsdfd_version", 0) + 1

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._bump()
        return r


class _VelocityNet(_Container):
    """cat[state, alpha, PE_P(omega_i)] -> n_hidden x (Linear no-bias, SiLU) -> Linear no-bias."""

    _N_HIDDEN = 0

    def __init__(self, input_dim=3, output_dim=1, N_NEURONS=32, POSITIONAL_ENCODING_BASIS_NUM=5):
        super().__init__()
        self.pos_num = POSITIONAL_ENCODING_BASIS_NUM
        self.input_dim = input_dim + 4 * POSITIONAL_ENCODING_BASIS_NUM
        widths = [self.input_dim] + [N_NEURONS] * self._N_HIDDEN
        for k in range(self._N_HIDDEN):
            setattr(self, f"linear{k + 1}", torch.nn.Linear(widths[k], widths[k + 1], bias=False))
        self.output = torch.nn.Linear(N_NEURONS, output_dim, bias=False)

    @property
    def n_hidden(self):
        return self._N_HIDDEN


class NN_cond_pos_simpler(_VelocityNet):
    _N_HIDDEN = 3


class NN_cond_pos(_VelocityNet):
    _N_HIDDEN = 4


class NN_cond_pos_spherical_complicate(_VelocityNet):
    _N_HIDDEN = 6

    def __init__(self, input_dim=3, output_dim=1, N_NEURONS=64, POSITIONAL_ENCODING_BASIS_NUM=5):
        super().__init__(input_dim, output_dim, N_NEURONS, POSITIONAL_ENCODING_BASIS_NUM)


class _BaseNet(_Container):
    """PE_P(omega_i) -> Linear(16)+b -> SiLU -> Linear(4)+b."""

    def __init__(self, input_dim, N_NEURONS, bands):
        super().__init__()
        self.POSITIONAL_ENCODING_BASIS_NUM = bands
        self.input_dim = input_dim + 4 * bands
        self.linear1 = torch.nn.Linear(self.input_dim, N_NEURONS)
        self.output = torch.nn.Linear(N_NEURONS, 4)


class NN_cond_pretrain_disk_one(_BaseNet):
    def __init__(self, input_dim=3, output_dim=4, N_NEURONS=16, POSITIONAL_ENCODING_BASIS_NUM=5):
        super().__init__(input_dim, N_NEURONS, POSITIONAL_ENCODING_BASIS_NUM)


class NN_cond_pretrain_spherical_one(_BaseNet):
    def __init__(self, input_dim=3, N_NEURONS=16, POSITIONAL_ENCODING_BASIS_NUM=3):
        super().__init__(input_dim, N_NEURONS, POSITIONAL_ENCODING_BASIS_NUM)
        self.n_modes = 1
        self.eps = 1e-3


def to_flow_weights(D_base: _BaseNet, D_sample: _VelocityNet, domain: int, name: str = "") -> W.FlowWeights:
    """Pull the fp32 weights of a reference-shaped (D_base, D_sample) pair."""
    sd = {k: v.detach().float().cpu().numpy() for k, v in D_sample.state_dict().items()}
    bd = {k: v.detach().float().cpu().numpy() for k, v in D_base.state_dict().items()}
    return W.from_state_dicts(name, domain, sd, bd, pe_bands=D_sample.pos_num,
                              base_pe_bands=D_base.POSITIONAL_ENCODING_BASIS_NUM)


def from_flow_weights(fw: W.FlowWeights):
    """Inverse of ``to_flow_weights``: containers filled from a ``.bsdfw`` weight set."""
    cls = {3: NN_cond_pos_simpler, 4: NN_cond_pos, 6: NN_cond_pos_spherical_complicate}.get(fw.n_hidden)
    if cls is None:
        raise ValueError(f"no reference container with {fw.n_hidden} hidden layers")
    ds = cls(input_dim=fw.state_dim + 3, output_dim=2, N_NEURONS=fw.width, POSITIONAL_ENCODING_BASIS_NUM=fw.pe_bands)
    bcls = NN_cond_pretrain_disk_one if fw.domain == W.DOMAIN_DISK else NN_cond_pretrain_spherical_one
    db = bcls(input_dim=2, N_NEURONS=fw.base_hidden, POSITIONAL_ENCODING_BASIS_NUM=fw.base_pe_bands)
    sd = {"linear1.weight": torch.from_numpy(fw.w_in.copy()), "output.weight": torch.from_numpy(fw.w_out.copy())}
    for k in range(fw.n_hidden - 1):
        sd[f"linear{k + 2}.weight"] = torch.from_numpy(fw.w_hidden[k].copy())
    ds.load_state_dict(sd)
    db.load_state_dict({"linear1.weight": torch.from_numpy(fw.base_w1.copy()),
                        "linear1.bias": torch.from_numpy(fw.base_b1.copy()),
                        "output.weight": torch.from_numpy(fw.base_w2.copy()),
                        "output.bias": torch.from_numpy(fw.base_b2.copy())})
    return db, ds
