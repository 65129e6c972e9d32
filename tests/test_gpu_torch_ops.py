"""The PyTorch-ROCm operator library used DIRECTLY (torch.ops.bsdfd.*, csrc/torch_ops.cpp), the way a reference
maintainer would after `torch.ops.load_library`: handles from a file and from in-memory nn.Linear-layout tensors, the
operators against the oracle, side streams, hipGraph capture, the C++ checks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from conftest import load_case, same_density  # noqa: E402
from oracle import bsdf_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    from bsdf_diffusion_sampling_amd import torch_ext
    return torch_ext.load()


def _dev():
    return torch.device("cuda", 0)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(_dev())


def test_handles_from_file_and_from_tensors_agree_with_the_oracle(ops):
    from bsdf_diffusion_sampling_amd import weights as W
    g, fw = load_case("chm_orange_rgb_spherical")
    T = int(g["meta_T"])
    h_file = ops.create_from_file(W.shipped_path("chm_orange_rgb", "spherical"), 0, 0)
    cpu = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))  # noqa: E731
    h_mem = ops.create(fw.domain, fw.width, fw.n_hidden, 0, cpu(fw.w_in), cpu(fw.w_hidden), cpu(fw.w_out), cpu(fw.base_w1),
                       cpu(fw.base_b1), cpu(fw.base_w2), cpu(fw.base_b2), 0)
    try:
        assert ops.flops_per_query(h_file, 8) == 165440
        wi, x0 = _t(g["wi"]), _t(g["x0"])
        xa, pa = ops.network_sampling(h_file, wi, x0, 0, 0, T)
        xb, pb = ops.network_sampling(h_mem, wi, x0, 0, 0, T)
        assert torch.equal(xa, xb) and torch.equal(pa, pb)
        xo, po = O.Oracle(fw).network_sampling(g["wi"], g["x0"], T)
        assert np.abs(xa.cpu().numpy() - xo).max() < 1e-4
        _, acc = O.Oracle(fw).flow(g["x0"], g["wi"], T, False)
        ok = (np.abs(acc) > 1e-3) & (np.abs(acc) < 1e3) & (np.abs(po) > 1e-6 * np.percentile(np.abs(po), 99))
        rel = np.abs(pa.cpu().numpy() - po)[ok] / np.abs(po[ok])
        assert np.percentile(rel, 99) < 1e-4
        pr = ops.network_pdf(h_file, xa, wi, T).cpu().numpy()
        pro = O.Oracle(fw).network_pdf(xa.cpu().numpy(), g["wi"], T)
        ok2 = ok & (np.abs(pro) > 1e-6 * np.percentile(np.abs(pro), 99))
        assert np.percentile(np.abs(pr - pro)[ok2] / np.abs(pro[ok2]), 99) < 1e-4
        # in-kernel draw: seeds are uint64 at the C ABI, int64 in the schema — the same 64 bits
        w1 = ops.plugin_sample(h_file, 0, _wi3(512), None, -1, 0, T)[0]
        w2 = ops.plugin_sample(h_file, 0, _wi3(512), None, -1, 0, T)[0]
        w3 = ops.plugin_sample(h_file, 0, _wi3(512), None, 12345, 0, T)[0]
        assert torch.equal(w1, w2) and not torch.equal(w1, w3)
    finally:
        ops.destroy(h_file)
        ops.destroy(h_mem)


def _wi3(n, seed=7):
    g = torch.Generator().manual_seed(seed)
    th, ph = 1.4 * torch.rand(n, generator=g), 6.28 * torch.rand(n, generator=g)
    return torch.stack([torch.sin(th) * torch.cos(ph), torch.sin(th) * torch.sin(ph), torch.cos(th)], 1).float().to(_dev())


def test_operators_run_on_the_current_stream_and_capture_into_a_graph(ops):
    from bsdf_diffusion_sampling_amd import weights as W
    h = ops.create_from_file(W.shipped_path("aniso_miro_7_rgb", "disk"), 0, 0)
    try:
        n = 8192
        wi = _wi3(n)
        x0 = (0.2 * torch.randn(n, 2, generator=torch.Generator().manual_seed(1))).to(_dev())
        ref_wo, ref_p = ops.plugin_sample(h, 0, wi, x0, 0, 0, 4)
        ref_pp = ops.plugin_pdf(h, 0, wi, ref_wo, 4)
        torch.cuda.synchronize()
        # a side stream: the operator must enqueue on it (a stale default-stream launch would race with the fill below)
        st = torch.cuda.Stream()
        wo = torch.empty_like(ref_wo)
        p = torch.empty_like(ref_p)
        with torch.cuda.stream(st):
            big = torch.zeros(64 << 20, device=_dev())          # keeps the side stream busy first
            big.add_(1.0)
            ops.plugin_sample_out(h, 0, wi, x0, 0, 0, 4, wo, p)
            pp = ops.plugin_pdf(h, 0, wi, wo, 4)
        st.synchronize()
        assert torch.equal(wo, ref_wo) and torch.equal(p, ref_p) and torch.equal(pp, ref_pp)
        # hipGraph capture + replay of sample -> pdf (nothing in the operators allocates from the host side of HIP or syncs)
        gph = torch.cuda.CUDAGraph()
        wo2, p2, pp2 = torch.empty_like(wo), torch.empty_like(p), torch.empty_like(p)
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            with torch.cuda.graph(gph, stream=cap):
                ops.plugin_sample_out(h, 0, wi, x0, 0, 0, 4, wo2, p2)
                ops.plugin_pdf_out(h, 0, wi, wo2, 4, pp2)
        wo2.zero_(); p2.zero_(); pp2.zero_()
        gph.replay()
        torch.cuda.synchronize()
        assert torch.equal(wo2, ref_wo) and torch.equal(p2, ref_p) and torch.equal(pp2, ref_pp)
        # fused sample + pdf of the same intersections
        wl = _wi3(n, 9)
        a, b, c = ops.plugin_sample_pdf(h, 0, wi, wl, x0, 0, 0, 4)
        assert torch.allclose(a, ref_wo, atol=2e-6) and same_density(c, ops.plugin_pdf(h, 0, wi, wl, 4))
    finally:
        ops.destroy(h)


def test_cxx_checks_throw(ops):
    from bsdf_diffusion_sampling_amd import weights as W
    h = ops.create_from_file(W.shipped_path("aniso_miro_7_rgb", "disk"), 0, 0)
    try:
        wi = _wi3(64)
        for bad in (wi.cpu(), wi.double(), wi[:, :2], wi.t().contiguous().t()):
            with pytest.raises(RuntimeError):
                ops.plugin_sample(h, 0, bad, None, 0, 0, 4)
        with pytest.raises(RuntimeError):
            ops.plugin_pdf(h, 0, wi, wi[:5].contiguous(), 4)                      # row count mismatch
        with pytest.raises(RuntimeError):
            ops.plugin_sample_out(h, 0, wi, None, 0, 0, 4, torch.empty_like(wi), torch.empty(63, device=_dev()))
        with pytest.raises(RuntimeError):
            ops.plugin_sample(h, 1, wi, None, 0, 0, 4)                            # full-sphere variant on a disk handle
        with pytest.raises(RuntimeError):
            ops.plugin_sample(h, 0, wi, None, 0, 0, 0)                            # T out of range
        with pytest.raises(RuntimeError):
            ops.plugin_sample(0, 0, wi, None, 0, 0, 4)                            # null handle
    finally:
        ops.destroy(h)
    with pytest.raises(RuntimeError):
        ops.create_from_file("/nonexistent/file.bsdfw", 0, 0)
