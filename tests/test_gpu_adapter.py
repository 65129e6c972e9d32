"""Control flow of mitsuba_adapter.py under a STUB `mitsuba` / `drjit` (neither is installable here: pip is index-less
and the offline wheelhouse has no such wheel).  The stub implements only what the adapter touches — Vector3f / Float /
UInt32 wrappers over torch tensors, BSDFSample3f, BSDF, load_dict, register_bsdf, dr.select — so this test pins the
adapter's OWN logic (conversions, field plumbing, delegation to the plugin cores), NOT compatibility with Mitsuba:
SURVEY §8 f3 stays "partial" until the adapter has run under the real renderer."""
import sys
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


class _Arr:
    def __init__(self, t):
        self.t = torch.as_tensor(t, dtype=torch.float32).cpu()

    def torch(self):
        return self.t


class _Vec3:
    def __init__(self, x, y=None, z=None):
        if y is None:  # Vector3f(0)
            self.x = self.y = self.z = _Arr(torch.zeros(1) + x)
        else:
            self.x, self.y, self.z = _Arr(x), _Arr(y), _Arr(z)

    def torch(self):
        return torch.stack([self.x.t, self.y.t, self.z.t], 1)


class _GT:  # a "measured" plugin: f cos = 0.3 * (1 + wi.x) * wo.z, grey
    def eval(self, ctx, si, wo):
        v = 0.3 * (1.0 + si.wi.x.t) * wo.z.t
        return _Vec3(v, v, v)


class _Props(dict):
    def has_property(self, k):
        return k in self


@pytest.fixture
def stub_mitsuba(monkeypatch):
    mi, dr = types.ModuleType("mitsuba"), types.ModuleType("drjit")
    registry = {}
    mi.variant = lambda: "llvm_ad_rgb"
    mi.set_variant = lambda v: None
    mi.Vector3f, mi.Float, mi.UInt32 = _Vec3, _Arr, _Arr

    class BSDF:
        def __init__(self, props):
            self.props = props
    mi.BSDF = BSDF
    mi.BSDFSample3f = lambda: types.SimpleNamespace()
    mi.load_dict = lambda d: _GT()
    mi.register_bsdf = lambda name, factory: registry.__setitem__(name, factory)
    mi.registry = registry

    def select(mask, a, b):
        if mask is True:
            return a
        m = torch.as_tensor(mask)
        return _Vec3(*(torch.where(m, ca.t, cb.t) for ca, cb in ((a.x, b.x), (a.y, b.y), (a.z, b.z))))
    dr.select = select
    monkeypatch.setitem(sys.modules, "mitsuba", mi)
    monkeypatch.setitem(sys.modules, "drjit", dr)
    return mi


def _wi(n, seed):
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n, 2, generator=g)
    r, a = 0.9 * torch.sqrt(u[:, 0]), 2 * np.pi * u[:, 1]
    x, y = r * torch.cos(a), r * torch.sin(a)
    return torch.stack([x, y, torch.sqrt(1 - x * x - y * y)], 1).float()


def test_adapter_delegates_to_the_plugin_core(stub_mitsuba):
    from bsdf_diffusion_sampling_amd import mitsuba_adapter as A
    from bsdf_diffusion_sampling_amd.brdf_measured_disk import MyBSDF as Core
    from bsdf_diffusion_sampling_amd.plugin_base import SurfaceInteraction
    mi = stub_mitsuba
    cls = A.register("mybsdf", domain="disk")
    plug = mi.registry["mybsdf"](_Props(filename="chm_orange_rgb", albedo=[0.8, 0.7, 0.6]))
    assert isinstance(plug, mi.BSDF) and plug.m_flags == plug.core.m_flags and type(plug) is cls
    n = 2048
    wi = _wi(n, 1)
    si = types.SimpleNamespace(wi=_Vec3(wi[:, 0], wi[:, 1], wi[:, 2]))
    dev = torch.device("cuda", 0)

    class TorchGT:  # the same ground truth, torch side
        def eval(self, ctx, s, wo):
            v = 0.3 * (1.0 + s.wi[:, 0]) * wo[:, 2]
            return torch.stack([v, v, v], 1)
    core = Core({"filename": "chm_orange_rgb", "albedo": [0.8, 0.7, 0.6], "bsdf": TorchGT()})
    # pdf / eval: identical to the core's
    wo = _wi(n, 2)
    p = plug.pdf(None, si, _Vec3(wo[:, 0], wo[:, 1], wo[:, 2])).torch()
    assert torch.equal(p, core.pdf(None, SurfaceInteraction(wi.to(dev)), wo.to(dev)).cpu())
    e = plug.eval(None, si, _Vec3(wo[:, 0], wo[:, 1], wo[:, 2])).torch()
    assert torch.allclose(e, core.eval(None, SurfaceInteraction(wi.to(dev)), wo.to(dev)).cpu(), rtol=1e-6, atol=1e-7)
    # sample: the Philox seed is drawn from torch's global generator (as the reference's torch.randn_like is), so the
    # same torch.manual_seed gives the adapter and a directly-called core the same draw — every field must be the core's
    torch.manual_seed(5)
    bs, weight = plug.sample(None, si, None, None)
    torch.manual_seed(5)
    bs_c, weight_c = core.sample(None, SurfaceInteraction(wi.to(dev)))
    w_o, pdf, wgt = bs.wo.torch(), bs.pdf.torch(), weight.torch()
    assert bs.eta == 1.0 and bs.sampled_type == core.m_flags and bs.sampled_component == 0
    assert torch.equal(w_o, bs_c.wo.cpu()) and torch.equal(pdf, bs_c.pdf.cpu())
    assert torch.allclose(wgt, weight_c.cpu(), rtol=1e-5, atol=1e-7)
    ok = pdf > 0
    assert ok.float().mean() > 0.5 and torch.allclose((w_o[ok] ** 2).sum(1), torch.ones(int(ok.sum())), atol=1e-4)
    # weight = f * albedo / pdf on kept lanes, zero elsewhere (firefly rule and masks are the core's)
    f = 0.3 * (1.0 + wi[:, 0]) * w_o[:, 2]
    expect = f[:, None] * torch.tensor([0.8, 0.7, 0.6]) / pdf[:, None]
    keep = ok & (w_o[:, 2] > 0) & (wi[:, 2] > 0)
    assert torch.allclose(wgt[keep], expect[keep], rtol=1e-4, atol=1e-6) and torch.count_nonzero(wgt[~keep]) == 0


def test_fullsphere_adapter_needs_a_ground_truth(stub_mitsuba):
    from bsdf_diffusion_sampling_amd import mitsuba_adapter as A
    cls = A.make_bsdf_class("fullsphere")
    with pytest.raises(RuntimeError, match="ground-truth"):
        cls(_Props(idx=3, albedo=[1, 1, 1]))
    plug = cls(_Props(idx=3, albedo=[1, 1, 1], bsdf=_GT()))
    n = 512
    g = torch.Generator().manual_seed(3)
    th, ph = 1.4 * torch.rand(n, generator=g), 6.28 * torch.rand(n, generator=g)
    wi = torch.stack([torch.sin(th) * torch.cos(ph), torch.sin(th) * torch.sin(ph), torch.cos(th)], 1)
    si = types.SimpleNamespace(wi=_Vec3(wi[:, 0], wi[:, 1], wi[:, 2]))
    bs, weight = plug.sample(None, si, None, None)
    up = bs.wo.torch()[:, 2] > 0
    # rendering/bsdf_myresult.py:89-90: eta 1 above / 1.788 below, sampled_type 8 / 16, component 2
    assert torch.equal(bs.eta.torch(), torch.where(up, 1.0, 1.788)) and bs.sampled_component == 2
    assert torch.equal(bs.sampled_type.torch(), torch.where(up, 8.0, 16.0))
    assert weight.torch().shape == (n, 3)
