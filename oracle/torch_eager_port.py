"""ORACLE — TEST INFRASTRUCTURE ONLY (CPU baseline leg).  Never imported by the product.

Op-for-op PyTorch-eager restatement of the reference's CPU path, used ONLY to time a
CPU baseline on the GPU box's host cores (``bench.py`` ``cpu_baseline``, kind "port"),
because the reference's own Python cannot travel there.  It executes
the same operator sequence per Euler step as the reference:

    PE(omega_i) recomputed inside every forward  (rendering/utils/model.py:494)
    cat -> Linear(no bias) -> SiLU chain          (model.py:490-501 / :435-446)
    two autograd backward() calls per step        (rendering/utils/mlp_brdf_sampling.py:33-41)
    x += v/T ; J rows ; tmp_J /= det              (:42-47)

and the base net is evaluated twice per sample() as the reference does (:20,:24).
``tests/test_oracle_golden.py`` checks its outputs against the goldens generated from
the reference, and ``tests/golden/make_golden.py --time`` recorded its wall time next to
the reference's on the same 8 cores.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn


def _pe(t, bands):
    parts = [t]
    for f in 2.0 ** torch.linspace(0.0, bands - 1, bands, dtype=t.dtype):
        parts += [torch.sin(t * f), torch.cos(t * f)]
    return torch.cat(parts, dim=-1)


class VelocityNet(nn.Module):
    def __init__(self, fw):
        super().__init__()
        self.bands = fw.pe_bands
        mats = [fw.w_in] + [w for w in fw.w_hidden] + [fw.w_out]
        self.layers = nn.ModuleList()
        for m in mats:
            lin = nn.Linear(m.shape[1], m.shape[0], bias=False)
            lin.weight.data = torch.from_numpy(m.copy())
            self.layers.append(lin)
        self.act = nn.SiLU()

    def forward(self, x, alpha, cond):
        h = torch.cat([x, alpha, _pe(cond, self.bands)], dim=1)
        for lin in self.layers[:-1]:
            h = self.act(lin(h))
        return self.layers[-1](h)


class BaseNet(nn.Module):
    def __init__(self, fw):
        super().__init__()
        self.bands = fw.base_pe_bands
        self.spherical = fw.domain == 1
        self.l1 = nn.Linear(fw.base_w1.shape[1], fw.base_w1.shape[0])
        self.l2 = nn.Linear(fw.base_w2.shape[1], 4)
        self.l1.weight.data, self.l1.bias.data = torch.from_numpy(fw.base_w1.copy()), torch.from_numpy(fw.base_b1.copy())
        self.l2.weight.data, self.l2.bias.data = torch.from_numpy(fw.base_w2.copy()), torch.from_numpy(fw.base_b2.copy())
        self.act, self.softplus = nn.SiLU(), nn.Softplus()

    def forward(self, cond):
        return self.l2(self.act(self.l1(_pe(cond, self.bands))))

    def sample(self, cond):
        o = self.forward(cond)
        if not self.spherical:
            return o[:, :2] + torch.randn_like(o[:, :2]) * torch.exp(o[:, 2:])
        th = o[:, :1] + torch.randn_like(o[:, :1]) * (torch.exp(o[:, 1:2]) + 1e-3)
        kap = self.softplus(o[:, 3]) + 1e-3
        ph = torch.distributions.von_mises.VonMises(o[:, 2], kap).sample().reshape(-1, 1)
        return torch.cat([th, ph], dim=1)

    def log_prob(self, x, cond):
        o = self.forward(cond)
        if not self.spherical:
            e = (x - o[:, :2]) / torch.exp(o[:, 2:])
            return -math.log(2 * math.pi) - o[:, 2:].sum(1) - 0.5 * (e ** 2).sum(1)
        e = (x[:, 0:1] - o[:, :1]) / (torch.exp(o[:, 1:2]) + 1e-3)
        lg = -0.5 * math.log(2 * math.pi) - o[:, 1:2].sum(1) - 0.5 * (e ** 2).sum(1)
        kap = self.softplus(o[:, 3]) + 1e-3
        return lg + torch.distributions.von_mises.VonMises(o[:, 2], kap).log_prob(x[:, 1])


def _step_inputs(x, spherical):
    if not spherical:
        return x
    return torch.cat([x[:, 0:1], torch.sin(x[:, 1:2]), torch.cos(x[:, 1:2])], dim=1)


def _flow(net, x, cond, T, reverse, spherical):
    n = x.shape[0]
    acc = torch.ones(n)
    e0 = torch.cat([torch.ones(n, 1), torch.zeros(n, 1)], 1)
    e1 = torch.cat([torch.zeros(n, 1), torch.ones(n, 1)], 1)
    sgn = -1.0 if reverse else 1.0
    x = x.detach().requires_grad_(True)
    for t in range(T):
        a = ((1 - t / T) if reverse else (t / T)) * torch.ones(n, 1)
        v = net(_step_inputs(x, spherical), a, cond)
        v.backward(e0, retain_graph=True)
        g0 = x.grad.clone()
        x.grad.zero_()
        v.backward(e1)
        g1 = x.grad.clone()
        r0 = e0 + sgn / T * g0
        r1 = e1 + sgn / T * g1
        det = r0[:, 0] * r1[:, 1] - r0[:, 1] * r1[:, 0]
        acc = acc * det if reverse else acc / det
        x = (x + sgn / T * v).detach().requires_grad_(True)
    return x.detach(), acc


def network_sampling(base, net, cond, T, x0=None):
    x0 = base.sample(cond) if x0 is None else x0
    p0 = base.log_prob(x0, cond).exp()
    x, acc = _flow(net, x0, cond, T, False, base.spherical)
    return x, (p0 * acc).detach()


def network_pdf(base, net, omega_o, cond, T):
    x, acc = _flow(net, omega_o.to(torch.float32), cond, T, True, base.spherical)
    with torch.no_grad():
        return base.log_prob(x, cond).exp() * acc
