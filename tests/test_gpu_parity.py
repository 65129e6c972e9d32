"""GPU parity tests proper: the HIP path (through the C ABI) vs the oracle and vs the
committed reference goldens, on the same seeded inputs.

Tolerances (BASELINE.json north_star: "within 1e-4 relative for sampled directions and
PDFs"; error metric of SURVEY.md §8(d)):
  * directions: absolute error <= 1e-4 on every component;
  * pdf: relative error over rows with |prod det J| > 1e-3 and a resolved density —
    p99 <= 1e-4 for precision "f32" and "split3" (the shipping default); sign must match;
  * precision "f16" (tcnn-class, tiny-cuda-nn/tmp.py:59 rtol=atol=1e-2): 1e-2 on directions.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from conftest import GOLDEN_CASES, LARGE_CASES, load_case, same_density  # noqa: E402
from oracle import bsdf_oracle as O  # noqa: E402


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    return torch.device("cuda", 0)


@pytest.fixture(autouse=True, params=["ctypes", "torch", "ctypes-tile16"])
def host_binding(request, monkeypatch):
    """Every parity test of this module runs through BOTH host shims over the C ABI: the ctypes binding and the
    PyTorch-ROCm operator library torch.ops.bsdfd.* (csrc/torch_ops.cpp) — and, third, with $BSDFD_TILE=16: the 16-query-tile
    kernels (csrc/bsdfd.hip) for the nets whose default is the 32-query-tile family (csrc/flow32.hip; bsdfd_desc.tile)."""
    binding, _, tile = request.param.partition("-tile")
    monkeypatch.setenv("BSDFD_HOST_BINDING", binding)
    if tile:
        monkeypatch.setenv("BSDFD_TILE", tile)
    else:
        monkeypatch.delenv("BSDFD_TILE", raising=False)
    yield binding


def _sampler(fw, precision):
    import os
    from bsdf_diffusion_sampling_amd.sampler import FlowSampler
    s = FlowSampler(fw, precision=precision)
    assert s.binding == os.environ["BSDFD_HOST_BINDING"]
    return s


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(_dev())


def _rel(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-30)


def _resolved(ref, acc):
    """Rows of the error metric: |prod det J| within [1e-3, 1e3] (the fp32
    reference itself loses digits where a step's det J ~ 0) and a density that is resolved
    relative to the bulk (99th percentile) of the batch."""
    det_ok = (np.abs(acc) > 1e-3) & (np.abs(acc) < 1e3)
    scale = np.percentile(np.abs(ref[det_ok]), 99)
    return det_ok & (np.abs(ref) > 1e-6 * scale)


def _record(name, **kv):
    """Achieved error figures, kept next to the run (gpurun_out/ is scratch; the committed copies are profiles/r0N_plugin_parity.json)."""
    import json
    import os
    from conftest import ROOT
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "plugin_parity.jsonl"), "a") as f:
            f.write(json.dumps({"test": name, **{k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in kv.items()}}) + "\n")
    except OSError:
        pass
    print(name, {k: (f"{v:.2e}" if isinstance(v, (float, np.floating)) else v) for k, v in kv.items()})


def _tail_bound(name, r, r32):
    """The p99 bounds above say nothing about the last 1 % of the rows: a defect confined to few rows (one lane of a cross-lane
    reduction on a partial tile, a rarely taken branch) would pass them.  Tail bound: the MAXIMUM relative error
    over the resolved rows is at most max(2e-3, 4 x the maximum of an fp32-arithmetic evaluation of the reference on the same
    rows) — near-singular steps lose digits in any fp32 evaluation, a wrong determinant is off by O(1).  The achieved figures
    are recorded (profiles/r0N_plugin_parity.json)."""
    bound = max(2e-3, 4.0 * float(r32.max()))
    _record(name, max=float(r.max()), fp32_reference_max=float(r32.max()), tail_bound=bound, p99=float(np.percentile(r, 99)))
    assert r.max() <= bound, (name, float(r.max()), bound)


@pytest.mark.parametrize("precision", ["f32", "split3"])
@pytest.mark.parametrize("stem", GOLDEN_CASES + LARGE_CASES)
def test_network_sampling_vs_oracle_and_golden(stem, precision):
    g, fw = load_case(stem)
    T = int(g["meta_T"])
    s = _sampler(fw, precision)
    x, p = s.network_sampling(_t(g["wi"]), _t(g["x0"]), T=T)
    x, p = x.cpu().numpy().astype(np.float64), p.cpu().numpy().astype(np.float64)
    orc = O.Oracle(fw)
    xo, po = orc.network_sampling(g["wi"], g["x0"], T)
    _, acc = orc.flow(g["x0"], g["wi"], T, reverse=False)
    assert np.abs(x - xo).max() < 1e-4
    ok = _resolved(po, acc)
    r = _rel(p, po)[ok]
    assert np.percentile(r, 99) < 1e-4, (np.median(r), np.percentile(r, 99), r.max())
    assert np.array_equal(np.sign(p[ok]), np.sign(po[ok]))
    # and against the reference's own fp32 outputs (golden)
    assert np.abs(x - g[f"sample_x_T{T}"]).max() < 1e-4
    rg = _rel(p, g[f"sample_pdf_T{T}"])[ok]
    assert np.percentile(rg, 99) < 2e-4
    # tail: against the reference's own fp32 run on the same rows (the golden IS that run)
    _tail_bound(f"network_sampling[{stem}:{precision}]", r, _rel(g[f"sample_pdf_T{T}"].astype(np.float64), po)[ok])


@pytest.mark.parametrize("precision", ["f32", "split3"])
@pytest.mark.parametrize("stem", GOLDEN_CASES + LARGE_CASES)
def test_network_pdf_vs_oracle_and_golden(stem, precision):
    g, fw = load_case(stem)
    s = _sampler(fw, precision)
    orc = O.Oracle(fw)
    for which in "ab":
        for T in (4, 8):
            key = f"pdf_{which}_T{T}"
            if key not in g.files:
                continue
            wo = g[f"pdf_wo_{which}"]
            p = s.network_pdf(_t(wo), _t(g["wi"]), T=T).cpu().numpy().astype(np.float64)
            po = orc.network_pdf(wo, g["wi"], T)
            _, acc = orc.flow(wo, g["wi"], T, reverse=True)
            ok = _resolved(po, acc)
            r = _rel(p, po)[ok]
            assert np.percentile(r, 99) < 1e-4, (which, T, np.median(r), np.percentile(r, 99), r.max())
            assert np.array_equal(np.sign(p[ok]), np.sign(po[ok]))
            rg = _rel(p, g[key])[ok]
            assert np.percentile(rg, 99) < 1e-3
            _tail_bound(f"network_pdf[{stem}:{precision}:{which}:T{T}]", r, _rel(g[key].astype(np.float64), po)[ok])

@pytest.mark.parametrize("T", [1, 2, 3, 5, 6, 7, 12, 33])
@pytest.mark.parametrize("stem", ["chm_orange_rgb_disk", "aniso_miro_7_rgb_spherical"])
def test_step_counts_incl_non_powers_of_two(stem, T):
    """T is an argument of the reference operators (mlp_brdf_sampling.py:17); alpha = t/T resp. 1 - t/T is
    formed in double and cast (the kernel has an exact-fp32 shortcut for powers of two and an fp64 path
    for the rest).  With T < 4 a single step's det(I + J/T) is ill-conditioned and fp32 arithmetic itself
    (precision "f32": exact fp32 FMA chains) exceeds 1e-4; there the bound is twice the f32 mode's error."""
    g, fw = load_case(stem)
    orc = O.Oracle(fw)
    xo, po = orc.network_sampling(g["wi"], g["x0"], T)
    _, acc = orc.flow(g["x0"], g["wi"], T, reverse=False)
    pro = orc.network_pdf(xo, g["wi"], T)
    _, accr = orc.flow(xo, g["wi"], T, reverse=True)
    ok, okr = _resolved(po, acc), _resolved(pro, accr)
    assert ok.sum() > 1000 and okr.sum() > 1000
    err = {}
    for prec in ("f32", "split3"):
        s = _sampler(fw, prec)
        x, p = s.network_sampling(_t(g["wi"]), _t(g["x0"]), T=T)
        x, p = x.cpu().numpy().astype(np.float64), p.cpu().numpy().astype(np.float64)
        assert np.abs(x - xo).max() < 1e-4
        pr = s.network_pdf(_t(xo), _t(g["wi"]), T=T).cpu().numpy().astype(np.float64)
        err[prec] = (np.percentile(_rel(p, po)[ok], 99), np.percentile(_rel(pr, pro)[okr], 99))
    for k in range(2):
        bound = 1e-4 if T >= 4 else max(1e-4, 2 * err["f32"][k])
        assert err["split3"][k] < bound, (T, k, err)


@pytest.mark.parametrize("stem", ["aniso_miro_7_rgb_disk", "aniso_miro_7_rgb_spherical",
                                  "aniso_miro_7_rgb_spherical_complex"])
def test_f16_precision_is_tcnn_class(stem):
    g, fw = load_case(stem)
    T = int(g["meta_T"])
    s = _sampler(fw, "f16")
    x, p = s.network_sampling(_t(g["wi"]), _t(g["x0"]), T=T)
    xo, po = O.Oracle(fw).network_sampling(g["wi"], g["x0"], T)
    assert np.abs(x.cpu().numpy() - xo).max() < 1e-2
    xs = s.flow_samples_only(_t(g["wi"]), _t(g["x0"]), T=T).cpu().numpy()
    # bsdfd_flow_samples_only in precision f16 is another fp16-class evaluation than network_sampling: its kernels (32-query tiles:
    # flow_kernel32w / flow_kernel32<.., SPLIT = false>, another summation order, state as hi + lo in layer 1; 16-query tiles: the
    # no-Jacobian instantiations) evaluate the hidden layers' sigmoids in PACKED fp16 on the pre-activation rounded to fp16
    # (csrc/flow_dev.h: act_pack8 — tiny-cuda-nn's FullyFusedMLP keeps its accumulators AND activations in fp16), the Jacobian
    # kernels in fp32.  Each inside the class: its own form of the bound, |x - oracle| <= atol + rtol |oracle| with rtol = atol =
    # 1e-2 (tiny-cuda-nn/tmp.py:59), on every row, and 99 % of the rows inside 5e-3
    err = np.abs(xs - xo)
    _record(f"f16_samples_only[{stem}:tile{s.tile_samples_only}]", p50=float(np.percentile(err, 50)), p99=float(np.percentile(err, 99)),
            max=float(err.max()), max_of_network_sampling=float(np.abs(x.cpu().numpy() - xo).max()))
    assert (err <= 1e-2 + 1e-2 * np.abs(xo)).all() and np.percentile(err, 99) < 5e-3, (float(np.percentile(err, 99)), float(err.max()))
    assert np.abs(xs - x.cpu().numpy()).max() < 2.5e-2


@pytest.mark.parametrize("stem", ["chm_orange_rgb_disk", "chm_orange_rgb_spherical"])
def test_flow_samples_only_matches_sampling_path(stem):
    g, fw = load_case(stem)
    s = _sampler(fw, "split3")
    for T in (1, 8, 32):
        x, _ = s.network_sampling(_t(g["wi"]), _t(g["x0"]), T=T)
        xs = s.flow_samples_only(_t(g["wi"]), _t(g["x0"]), T=T)
        assert torch.allclose(x, xs, atol=1e-6, rtol=0)


@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 1000, 4097])
def test_ragged_sizes(n):
    g, fw = load_case("chm_orange_rgb_disk")
    s = _sampler(fw, "split3")
    wi, x0 = g["wi"][:n].reshape(-1, 2) if n <= 2048 else np.tile(g["wi"], (3, 1))[:n], None
    x0 = g["x0"][:n].reshape(-1, 2) if n <= 2048 else np.tile(g["x0"], (3, 1))[:n]
    x, p = s.network_sampling(_t(wi), _t(x0), T=4)
    assert x.shape == (n, 2) and p.shape == (n,)
    if n:
        xo, po = O.Oracle(fw).network_sampling(wi, x0, 4)
        assert np.abs(x.cpu().numpy() - xo).max() < 1e-4
        # a canary after the buffer must be untouched (tail handled in-kernel)
        big = torch.full((n + 16,), -7.0, device=_dev())
        s._L.bsdfd_network_pdf(s._h, _t(xo).data_ptr(), _t(wi).data_ptr(), n, 4, big.data_ptr(), None)
        torch.cuda.synchronize()
        assert torch.all(big[n:] == -7.0)


@pytest.mark.parametrize("n", [1, 7, 17, 1000])
def test_spherical_plugin_ragged_tile_equals_the_padded_launch(n):
    """The spherical plugins deal the atan2f evaluations of cart_to_spher to the four lanes of a query and hand the results
    round by lane shuffles (csrc/bsdfd.hip, atan2_by_lane): a partial last tile must give, row for row, what the same rows
    give inside a full launch - sample(), pdf() without and with the per-query context, and the fused call."""
    g, fw = load_case("chm_orange_rgb_spherical")
    s = _sampler(fw, "split3")
    rng = np.random.default_rng(n)
    full = 1024 + 16
    wi3 = _dir(rng.uniform(0.01, 1.5, full), rng.uniform(-np.pi, np.pi, full))
    wl3 = _dir(rng.uniform(0.01, 1.5, full), rng.uniform(-np.pi, np.pi, full))
    x0 = np.tile(g["x0"], (2, 1))[:full]
    wo_f, p_f = s.plugin_sample(_t(wi3), _t(x0), T=8)
    q_f = s.plugin_pdf(_t(wi3), _t(wl3), T=8)
    wo_n, p_n = s.plugin_sample(_t(wi3[:n]), _t(x0[:n]), T=8)
    ctx = s.new_context(n)
    q_n = s.plugin_pdf(_t(wi3[:n]), _t(wl3[:n]), T=8, ctx_out=ctx)
    q_c = s.plugin_pdf(_t(wi3[:n]), _t(wl3[:n]), T=8, ctx_in=ctx)
    assert torch.equal(wo_n, wo_f[:n]) and torch.equal(p_n, p_f[:n])
    assert torch.equal(q_n, q_f[:n]) and torch.equal(q_c, q_f[:n])
    wo_u, p_u, q_u = s.plugin_sample_pdf(_t(wi3[:n]), _t(wl3[:n]), _t(x0[:n]), T=8)
    wo_g, p_g, q_g = s.plugin_sample_pdf(_t(wi3), _t(wl3), _t(x0), T=8)
    assert torch.equal(wo_u, wo_g[:n]) and torch.equal(p_u, p_g[:n]) and torch.equal(q_u, q_g[:n])
    assert torch.isfinite(q_n).all() and (q_n > 0).any()


def _wi3_disk(wi2):
    return np.concatenate([wi2, np.sqrt(np.maximum(1 - (wi2 ** 2).sum(1), 0))[:, None]], 1).astype(np.float32)


def _dir(th, ph):
    return np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)], 1).astype(np.float32)


def test_plugin_disk_sample_and_pdf():
    g, fw = load_case("aniso_miro_7_rgb_disk")
    s = _sampler(fw, "split3")
    orc = O.Oracle(fw)
    wi3 = _wi3_disk(g["wi"])
    x0 = g["x0"].copy()
    x0[:16] *= 40.0  # push some samples out of the disk: r^2 >= 0.995 guard
    wo, pdf = s.plugin_sample(_t(wi3), _t(x0), T=4)
    wo, pdf = wo.cpu().numpy(), pdf.cpu().numpy()
    wo_o, pdf_o = O.plugin_sample_disk(orc, wi3, x0, T=4)
    assert np.abs(wo - wo_o).max() < 1e-4
    bad = (wo_o[:, 0] == 0) & (wo_o[:, 1] == 0)
    assert bad.sum() >= 1 and np.all(pdf[bad] == 0) and np.all(wo[bad] == np.array([0, 0, 1], np.float32))
    ok = ~bad & (np.abs(pdf_o) > 1e-6 * np.abs(pdf_o).max())
    assert np.percentile(_rel(pdf, pdf_o)[ok], 99) < 2e-4
    # pdf(): cos masks
    wo3 = wo_o.astype(np.float32).copy()
    wo3[:8, 2] *= -1
    wi3m = wi3.copy()
    wi3m[8:16, 2] *= -1
    p = s.plugin_pdf(_t(wi3m), _t(wo3), T=4).cpu().numpy()
    p_o = O.plugin_pdf_disk(orc, wi3m, wo3, T=4)
    assert np.all(p[:16] == 0) and np.all(p_o[:16] == 0)
    ok = np.abs(p_o) > 1e-6 * np.abs(p_o).max()
    assert np.percentile(_rel(p, p_o)[ok], 99) < 2e-4


@pytest.mark.parametrize("stem,full", [("aniso_miro_7_rgb_spherical", False), ("bsdf_3_spherical", True)])
def test_plugin_spherical_sample_and_pdf(stem, full):
    from bsdf_diffusion_sampling_amd import _lib
    g, fw = load_case(stem)
    s = _sampler(fw, "split3")
    orc = O.Oracle(fw)
    variant = _lib.PLUGIN_FULLSPHERE if full else _lib.PLUGIN_MEASURED
    # (down to 1e-3 rad of the pole: the kernel evaluates cart_to_spher in its well-conditioned form, csrc/bsdfd.hip; the fp32
    #  restatement below shows what acos() as written loses there)
    th = np.clip(g["wi"][:, 0].astype(np.float64), 0.001, None)
    wi3 = _dir(th, g["wi"][:, 1].astype(np.float64))
    wo, pdf = s.plugin_sample(_t(wi3), _t(g["x0"]), T=8, variant=variant)
    wo, pdf = wo.cpu().numpy(), pdf.cpu().numpy()
    wo_o, pdf_o = O.plugin_sample_spherical(orc, wi3.astype(np.float64), g["x0"], T=8, full_sphere=full)
    # the same restatement in fp32 arithmetic = the noise a fp32 implementation of these lines carries on these rows
    # (rendering/brdf_measured_spherical.py:35-39: acos / atan2 of the unit vector; :30-33 sincos back)
    orc32 = O.Oracle(fw, np.float32)
    wo_32, pdf_32 = O.plugin_sample_spherical(orc32, wi3, g["x0"], T=8, full_sphere=full)
    err_wo, noise_wo = np.abs(wo - wo_o), np.abs(wo_32.astype(np.float64) - wo_o)
    assert np.percentile(err_wo, 99) <= 1e-5 and err_wo.max() <= 1e-4
    assert np.allclose((wo ** 2).sum(1), 1.0, atol=1e-5)
    if not full:
        assert np.all(pdf[wo_o[:, 2] < -1e-4] == 0)
    _, acc = orc.flow(g["x0"], O.cart_to_spher(wi3.astype(np.float64)), 8, reverse=False)
    ok = _resolved(pdf_o, acc)
    e, n32 = _rel(pdf, pdf_o)[ok], _rel(pdf_32.astype(np.float64), pdf_o)[ok]
    _record(f"plugin_spherical_sample[{stem}]", wo_p99=np.percentile(err_wo, 99), wo_max=err_wo.max(), pdf_median=np.median(e),
            pdf_p99=np.percentile(e, 99), fp32_oracle_wo_max=noise_wo.max(), fp32_oracle_pdf_p99=np.percentile(n32, 99))
    assert np.percentile(e, 99) <= 1e-4
    # pdf() on fresh directions
    tho = np.clip(g["pdf_wo_b"][:, 0].astype(np.float64), 0.05, 3.09)
    wo3 = _dir(tho, g["pdf_wo_b"][:, 1].astype(np.float64))
    p = s.plugin_pdf(_t(wi3), _t(wo3), T=8, variant=variant).cpu().numpy()
    p_o = O.plugin_pdf_spherical(orc, wi3.astype(np.float64), wo3.astype(np.float64), T=8, full_sphere=full)
    p_32 = O.plugin_pdf_spherical(orc32, wi3, wo3, T=8, full_sphere=full)
    _, acc = orc.flow(O.cart_to_spher(wo3.astype(np.float64)), O.cart_to_spher(wi3.astype(np.float64)), 8, reverse=True)
    ok = _resolved(p_o, acc)
    e, n32 = _rel(p, p_o)[ok], _rel(p_32.astype(np.float64), p_o)[ok]
    _record(f"plugin_spherical_pdf[{stem}]", pdf_median=np.median(e), pdf_p99=np.percentile(e, 99), fp32_oracle_pdf_p99=np.percentile(n32, 99))
    assert np.percentile(e, 99) <= 1e-4
    if not full:
        assert np.all(p[wo3[:, 2] <= 0] == 0)


def test_spherical_directions_at_and_near_the_pole():
    """cart_to_spher (rendering/brdf_measured_spherical.py:35-39) for wi and wo within 1e-6 .. 1e-2 rad of the normal, and exactly
    on it: acos(z / (r + 1e-8)) as written loses the angle there in fp32 (the quotient rounds to 1: theta = 0, or to the next
    float: 3.5e-4); the kernel's form of the same angle follows the fp64 oracle to the usual bounds.  pdf() keeps the
    reference's guard decision on the axis (fp32 theta_o = 0 -> sin(theta_o) > 5e-5 fails -> 0) for the measured variant."""
    from bsdf_diffusion_sampling_amd import _lib
    g, fw = load_case("chm_orange_rgb_spherical")
    s = _sampler(fw, "split3")
    orc, orc32 = O.Oracle(fw), O.Oracle(fw, np.float32)
    rng = np.random.default_rng(11)
    n = g["x0"].shape[0]
    th = 10.0 ** rng.uniform(-6, -2, n)
    th[:4] = 0.0                                                     # exactly the normal
    wi3 = _dir(th, rng.uniform(-np.pi, np.pi, n))
    wo, pdf = s.plugin_sample(_t(wi3), _t(g["x0"]), T=8)
    wo, pdf = wo.cpu().numpy(), pdf.cpu().numpy()
    wo_o, pdf_o = O.plugin_sample_spherical(orc, wi3.astype(np.float64), g["x0"], T=8)
    wo_32, pdf_32 = O.plugin_sample_spherical(orc32, wi3, g["x0"], T=8)
    _, acc = orc.flow(g["x0"], O.cart_to_spher(wi3.astype(np.float64)), 8, reverse=False)
    ok = _resolved(pdf_o, acc)
    e, n32 = _rel(pdf, pdf_o)[ok], _rel(pdf_32.astype(np.float64), pdf_o)[ok]
    _record("plugin_spherical_sample_near_pole[chm_orange_rgb_spherical]", wo_max=np.abs(wo - wo_o).max(), pdf_p99=np.percentile(e, 99),
            fp32_oracle_wo_max=np.abs(wo_32 - wo_o).max(), fp32_oracle_pdf_p99=np.percentile(n32, 99))
    assert np.isfinite(wo).all() and np.isfinite(pdf).all()
    assert np.abs(wo - wo_o).max() <= 1e-4 and np.percentile(e, 99) <= 1e-4
    # pdf(): wo near the pole (the 1 / sin(theta_o) Jacobian is clamped as in the reference), wi ordinary
    wi_b = _dir(np.clip(g["wi"][:, 0].astype(np.float64), 0.05, 1.5), g["wi"][:, 1].astype(np.float64))
    tho = 10.0 ** rng.uniform(-3.3, -2, n)
    wo3 = _dir(tho, rng.uniform(-np.pi, np.pi, n))
    p = s.plugin_pdf(_t(wi_b), _t(wo3), T=8).cpu().numpy()
    p_o = O.plugin_pdf_spherical(orc, wi_b.astype(np.float64), wo3.astype(np.float64), T=8)
    _, acc = orc.flow(O.cart_to_spher(wo3.astype(np.float64)), O.cart_to_spher(wi_b.astype(np.float64)), 8, reverse=True)
    ok = _resolved(p_o, acc)
    e = _rel(p, p_o)[ok]
    _record("plugin_spherical_pdf_near_pole[chm_orange_rgb_spherical]", pdf_p99=np.percentile(e, 99))
    assert np.percentile(e, 99) <= 1e-4
    axis = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (8, 1))
    assert np.all(s.plugin_pdf(_t(wi_b[:8]), _t(axis), T=8).cpu().numpy() == 0)
    pf = s.plugin_pdf(_t(wi_b[:8]), _t(axis), T=8, variant=_lib.PLUGIN_FULLSPHERE).cpu().numpy()
    assert not np.isnan(pf).any() and np.all(pf > 0)   # (density x the clamped 1 / sin(theta_o) = 3.4e38)


@pytest.mark.parametrize("stem", ["chm_orange_rgb_disk", "aniso_miro_7_rgb_spherical", "bsdf_3_spherical"])
def test_in_kernel_rng_statistics(stem):
    """x0 drawn in-kernel (Philox): only statistical parity with torch's RNG stream is
    attainable.  With the output layer zeroed the flow is the identity, so
    the call returns the base draw itself and its density."""
    import copy
    from scipy import stats
    g, fw = load_case(stem)
    fz = copy.deepcopy(fw)
    fz.w_out = np.zeros_like(fw.w_out)
    s = _sampler(fz, "split3")
    orc = O.Oracle(fz)
    n = 1 << 17
    for row in (0, 5):
        wi = np.tile(g["wi"][row:row + 1], (n, 1))
        x1, p1 = s.network_sampling(_t(wi), None, T=2, seed=7)
        x2, _ = s.network_sampling(_t(wi), None, T=2, seed=7)
        x3, _ = s.network_sampling(_t(wi), None, T=2, seed=8)
        x4, _ = s.network_sampling(_t(wi), None, T=2, seed=7, offset=n)
        assert torch.equal(x1, x2) and not torch.equal(x1, x3) and not torch.equal(x1, x4)
        x, p = x1.cpu().numpy().astype(np.float64), p1.cpu().numpy().astype(np.float64)
        po = np.exp(orc.base_log_prob(x, wi))
        ok = po > 1e-6 * po.max()
        assert np.percentile(_rel(p, po)[ok], 99) < 1e-4
        o = orc.base_forward(wi[:1])[0]
        if fw.domain == O.DOMAIN_DISK:
            for d in (0, 1):
                sd = np.exp(o[2 + d])
                assert abs(x[:, d].mean() - o[d]) < 5 * sd / np.sqrt(n)
                assert abs(x[:, d].std() / sd - 1) < 0.02
                assert stats.kstest((x[:, d] - o[d]) / sd, "norm").pvalue > 1e-4
            assert abs(np.corrcoef(x[:, 0], x[:, 1])[0, 1]) < 0.02
        else:
            sd = np.exp(o[1]) + 1e-3
            assert stats.kstest((x[:, 0] - o[0]) / sd, "norm").pvalue > 1e-4
            mu, kappa = orc.base_von_mises_params(wi[:1])
            assert np.all(x[:, 1] >= -np.pi - 1e-6) and np.all(x[:, 1] <= np.pi + 1e-6)
            d = np.angle(np.exp(1j * (x[:, 1] - mu[0])))
            assert stats.kstest(d, stats.vonmises(kappa[0]).cdf).pvalue > 1e-4


def test_error_paths():
    from bsdf_diffusion_sampling_amd.sampler import FlowSampler
    g, fw = load_case("chm_orange_rgb_disk")
    s = FlowSampler(fw)
    wi = _t(g["wi"])
    with pytest.raises(RuntimeError):
        s.network_sampling(wi.cpu(), None)
    with pytest.raises(RuntimeError):
        s.network_sampling(wi.double(), None)
    with pytest.raises(RuntimeError):
        s.network_sampling(wi.t().contiguous().t(), None) if False else s.network_sampling(wi[:, :1], None)
    with pytest.raises(RuntimeError):
        s.network_pdf(wi[:5], wi)
    with pytest.raises(RuntimeError):
        s.network_sampling(wi, None, T=0)
    with pytest.raises(RuntimeError):
        s.plugin_sample(torch.zeros(4, 3, device=_dev()), None, variant=1)  # full sphere on a disk handle


@pytest.mark.parametrize("raw_kappa", [-8.0, -0.5, 3.70, 3.80, 25.0, 400.0])
def test_base_density_kappa_branches_and_von_mises_sampler(raw_kappa):
    """Crafted weights: base output = its bias (constant loc/log_scale/mu/kappa_raw) and a zero
    output layer (identity flow), so pdf == base density and samples == base draws.  Sweeps kappa
    across softplus' threshold-free range and log I0's 3.75 polynomial switch (model.py:294-317,
    torch von_mises._log_modified_bessel_fn), incl. the tiny-kappa proposal (fp64 in-kernel)."""
    import copy
    from scipy import stats
    g, fw = load_case("aniso_miro_7_rgb_spherical")
    fz = copy.deepcopy(fw)
    fz.w_out = np.zeros_like(fw.w_out)
    fz.base_w2 = np.zeros_like(fw.base_w2)
    fz.base_b2 = np.array([0.7, -1.2, 0.4, raw_kappa], np.float32)
    s = _sampler(fz, "split3")
    orc = O.Oracle(fz)
    n = 1 << 16
    wi = np.tile(g["wi"][:1], (n, 1))
    rng = np.random.default_rng(3)
    x = np.stack([rng.normal(0.7, 0.4, n), rng.uniform(-np.pi, np.pi, n)], 1).astype(np.float32)
    p = s.network_pdf(_t(x), _t(wi), T=2).cpu().numpy().astype(np.float64)
    po = np.exp(orc.base_log_prob(x, wi))
    ok = po > 1e-8 * po.max()
    assert np.percentile(_rel(p, po)[ok], 99) < 2e-5 + 3e-7 * max(1.0, np.log1p(np.exp(raw_kappa)))
    xs, ps = s.network_sampling(_t(wi), None, T=2, seed=11)
    xs = xs.cpu().numpy().astype(np.float64)
    mu, kappa = orc.base_von_mises_params(wi[:1])
    assert np.all(np.abs(xs[:, 1]) <= np.pi + 1e-5)
    d = np.angle(np.exp(1j * (xs[:, 1] - mu[0])))
    assert stats.kstest(d, stats.vonmises(kappa[0]).cdf).pvalue > 1e-4
    sd = np.exp(-1.2) + 1e-3
    assert stats.kstest((xs[:, 0] - 0.7) / sd, "norm").pvalue > 1e-4
    pso = np.exp(orc.base_log_prob(xs, wi))
    ok = pso > 1e-8 * pso.max()
    assert np.percentile(_rel(ps.cpu().numpy(), pso)[ok], 99) < 1e-4


def test_guards_spherical_pole_and_horizon():
    """sin(theta_o) <= 5e-5 and cos(theta_o) <= 0 guards of brdf_measured_spherical.py:79-80,134 with
    crafted states: identity flow (zero output layer) returns x0 itself."""
    import copy
    from bsdf_diffusion_sampling_amd import _lib
    g, fw = load_case("aniso_miro_7_rgb_spherical")
    fz = copy.deepcopy(fw)
    fz.w_out = np.zeros_like(fw.w_out)
    fz.base_w2 = np.zeros_like(fw.base_w2)           # a broad, constant base density: nothing underflows
    fz.base_b2 = np.array([1.0, 0.5, 0.0, 0.0], np.float32)
    s = _sampler(fz, "split3")
    wi3 = _dir(np.full(8, 0.6), np.linspace(-3, 3, 8))
    x0 = np.array([[1e-5, 0.3], [4.9e-5, 1.0], [6e-5, 1.0], [1.0, 2.0], [1.5707, -1.0], [1.5709, 0.5],
                   [2.5, 0.1], [3.14159, 0.0]], np.float32)
    wo, pdf = s.plugin_sample(_t(wi3), _t(x0), T=4)
    wo, pdf = wo.cpu().numpy(), pdf.cpu().numpy()
    assert np.all(pdf[[0, 1]] == 0) and pdf[2] != 0 and pdf[3] != 0      # sin guard
    assert pdf[4] != 0 and np.all(pdf[[5, 6, 7]] == 0)                     # cos guard (hemisphere)
    assert np.allclose(wo[3], [np.cos(2.0) * np.sin(1.0), np.sin(2.0) * np.sin(1.0), np.cos(1.0)], atol=1e-6)
    wo_f, pdf_f = s.plugin_sample(_t(wi3), _t(x0), T=4, variant=_lib.PLUGIN_FULLSPHERE)
    pdf_f = pdf_f.cpu().numpy()
    assert np.all(pdf_f[[0, 1]] == 0) and np.all(pdf_f[[2, 3, 4, 5, 6]] != 0)  # no cos guard on the full sphere
    assert np.isfinite(pdf_f).all()
    # pdf(): lanes with cos(theta_i) <= 0 or cos(theta_o) <= 0 are zero for the measured variant only
    wo3 = _dir(np.array([0.5, 2.0, 0.5, 0.5]), np.zeros(4))
    wi4 = _dir(np.array([0.5, 0.5, 2.0, 0.5]), np.ones(4))
    pm = s.plugin_pdf(_t(wi4), _t(wo3), T=4).cpu().numpy()
    pf = s.plugin_pdf(_t(wi4), _t(wo3), T=4, variant=_lib.PLUGIN_FULLSPHERE).cpu().numpy()
    assert pm[0] != 0 and pm[1] == 0 and pm[2] == 0 and np.all(pf != 0)


def test_reflow_teacher_sampler_T128_fp16_class():
    """f2: the reference's only tiny-cuda-nn call site — 64-wide x 6 teacher, T = 128 Euler steps, no
    Jacobian (learning_repo_cleanup/spherical_domain_sampling.py:147-166).  Its own accepted tolerance
    for the fp16 FullyFusedMLP is rtol = atol = 1e-2 per network evaluation (tiny-cuda-nn/tmp.py:59);
    over 128 steps we require the fp16 path to stay within 2e-2 of the fp64 oracle on >= 99 % of the
    rows, and the split3 path within 1e-4 on all."""
    g, fw = load_case("aniso_miro_7_rgb_spherical_complex")
    n, T = 1024, 128
    wi, x0 = g["wi"][:n], g["x0"][:n]
    xo, _ = O.Oracle(fw).flow(x0, wi, T, reverse=False)
    import os
    s16 = _sampler(fw, "f16")
    # the fp16 teacher has a 32-query-tile kernel of its own for this call (csrc/flow32.hip: flow_kernel32w) — the default
    assert s16.tile_samples_only == (16 if os.environ.get("BSDFD_TILE") == "16" else 32) and s16.tile == 16
    x16 = s16.flow_samples_only(_t(wi), _t(x0), T=T).cpu().numpy()
    xs3 = _sampler(fw, "split3").flow_samples_only(_t(wi), _t(x0), T=T).cpu().numpy()
    e16 = np.abs(x16 - xo).max(1)
    assert np.percentile(e16, 99) < 2e-2, np.percentile(e16, [50, 99, 100])
    assert np.abs(xs3 - xo).max() < 1e-4, np.abs(xs3 - xo).max()
    # ragged N / a query count that is not a multiple of the tile: same rows, same values; non-power-of-two T
    for m in (1, 31, 333):
        assert np.array_equal(s16.flow_samples_only(_t(wi[:m]), _t(x0[:m]), T=T).cpu().numpy(), x16[:m])
    xo7, _ = O.Oracle(fw).flow(x0[:256], wi[:256], 7, reverse=False)
    e7 = np.abs(s16.flow_samples_only(_t(wi[:256]), _t(x0[:256]), T=7).cpu().numpy() - xo7).max(1)
    assert np.percentile(e7, 99) < 2e-2, np.percentile(e7, [50, 99, 100])


@pytest.mark.parametrize("T", [128, 256])
def test_disk_reflow_teacher_sampler_long_T(T):
    """f2, DISK teacher (learning_repo_cleanup/disk_domain_sampling.py:93-110): the 32 x 3 `brdf_diffusion_network`,
    T = 128 and the script's default T = 256 Euler steps, no Jacobian.  HIP samples-only path vs the fp64 oracle AND
    vs the reference's own module stepped in fp32 (tests/golden/make_teacher_golden.py): split3 within 1e-4 on every
    row, the fp16 (tiny-cuda-nn class, tmp.py:59 rtol = atol = 1e-2) path within 2e-2 on >= 99 % of the rows."""
    import os
    from conftest import GOLDEN
    from bsdf_diffusion_sampling_amd import weights as W
    g = np.load(os.path.join(GOLDEN, "disk_teacher_aniso_miro_7_rgb.npz"))
    fw = W.load(W.shipped_path("aniso_miro_7_rgb", "disk", "diffusion"))
    wi, x0 = g["wi"], g["x0"]
    xo, _ = O.Oracle(fw).flow(x0, wi, T, reverse=False)
    xs3 = _sampler(fw, "split3").flow_samples_only(_t(wi), _t(x0), T=T).cpu().numpy()
    assert np.abs(xs3 - xo).max() < 1e-4, np.abs(xs3 - xo).max()
    assert np.abs(xs3 - g[f"x_T{T}_f32"]).max() < 1e-4
    x16 = _sampler(fw, "f16").flow_samples_only(_t(wi), _t(x0), T=T).cpu().numpy()
    e16 = np.abs(x16 - xo).max(1)
    assert np.percentile(e16, 99) < 2e-2, np.percentile(e16, [50, 99, 100])
    # ragged N and a query count that is not a multiple of the tile: same rows, same values
    x_part = _sampler(fw, "split3").flow_samples_only(_t(wi[:333]), _t(x0[:333]), T=T).cpu().numpy()
    assert np.array_equal(x_part, xs3[:333])


def test_operator_functions_repack_after_in_place_weight_updates():
    """The four operator functions cache the packed device handle on D_sample; the cache key carries data_ptr and
    torch's version counter of every weight, so an optimiser-style in-place update (the reflow training loop samples
    from the net it is updating, learning_repo_cleanup/disk_domain_sampling.py:112-140) is never served stale weights."""
    from bsdf_diffusion_sampling_amd import mlp_brdf_sampling as ops
    from bsdf_diffusion_sampling_amd import model as M
    from bsdf_diffusion_sampling_amd import weights as W
    g, fw = load_case("chm_orange_rgb_disk")
    db, ds = M.from_flow_weights(fw)
    wi, x0 = _t(g["wi"][:512]), _t(g["x0"][:512])
    x_a, p_a = ops.network_sampling_disk(db, ds, wi, T=4, x0=x0)
    s_a = ds.__dict__["_bsdfd_cache"]
    x_b, _ = ops.network_sampling_disk(db, ds, wi, T=4, x0=x0)
    assert torch.equal(x_a, x_b) and ds.__dict__["_bsdfd_cache"] is s_a and len(s_a) == 1   # cached: no re-pack
    with torch.no_grad():
        ds.output.weight.mul_(0.5)                                                           # what an optimiser step does
    x_c, _ = ops.network_sampling_disk(db, ds, wi, T=4, x0=x0)
    fw2 = M.to_flow_weights(db, ds, W.DOMAIN_DISK)
    xo, _ = O.Oracle(fw2).network_sampling(g["wi"][:512], g["x0"][:512], 4)
    assert not torch.equal(x_a, x_c) and np.abs(x_c.cpu().numpy() - xo).max() < 1e-4
    with torch.no_grad():
        db.output.bias.add_(0.01)                                                            # the base net counts too
    _, p_d = ops.network_sampling_disk(db, ds, wi, T=4, x0=x0)
    _, p_c = ops.network_sampling_disk(db, ds, wi, T=4, x0=x0)
    assert torch.equal(p_c, p_d)
    ds.output.weight.data.mul_(2.0)                                                          # invisible to the counter ...
    ops.repack(ds)                                                                           # ... hence the explicit hook
    x_e, _ = ops.network_sampling_disk(db, ds, wi, T=4, x0=x0)
    assert torch.allclose(x_e, x_a, atol=1e-6)


def test_extreme_inputs_do_not_crash_and_fail_loudly():
    """Absurd states (|x0| = 1e4 .. 1e30, NaN): the fp32 reference overflows its own exp()/products; the
    split-fp16 path additionally leaves fp16 range.  Required: no fault, the neighbouring queries of the
    same 16-query tile are untouched, and an out-of-range query reports a non-finite or zero pdf rather
    than a plausible finite one."""
    g, fw = load_case("chm_orange_rgb_disk")
    s = _sampler(fw, "split3")
    wi, x0 = g["wi"][:64].copy(), g["x0"][:64].copy()
    ref_x, ref_p = s.network_sampling(_t(wi), _t(x0), T=4)
    bad = x0.copy()
    bad[3] = [1e4, -1e4]
    bad[17] = [1e30, 1e30]
    bad[40] = [np.nan, 0.1]
    x, p = s.network_sampling(_t(wi), _t(bad), T=4)
    keep = np.ones(64, bool)
    keep[[3, 17, 40]] = False
    assert torch.equal(x[keep], ref_x[keep]) and torch.equal(p[keep], ref_p[keep])
    pb = p.cpu().numpy()[[3, 17, 40]]
    assert np.all(~np.isfinite(pb) | (pb == 0))


@pytest.mark.parametrize("stem,variant", [("chm_orange_rgb_disk", 0), ("aniso_miro_7_rgb_spherical", 0), ("bsdf_3_spherical", 1),
                                          ("aniso_miro_7_rgb_spherical_complex", 0)])
def test_fused_sample_pdf_equals_the_two_calls(stem, variant):
    """bsdfd_plugin_sample_pdf = plugin_sample(wi) + plugin_pdf(wi, wl) on one evaluation of the prologue."""
    g, fw = load_case(stem)
    s = _sampler(fw, "split3")
    rng = np.random.default_rng(0)
    n = 30001
    def dirs(lo):
        z, ph = rng.uniform(lo, 1.0, size=n), rng.uniform(0, 2 * np.pi, size=n)
        r = np.sqrt(1 - z * z)
        return _t(np.stack([r * np.cos(ph), r * np.sin(ph), z], 1))
    wi, wl = dirs(0.05), dirs(-1.0 if variant else 0.02)
    T = 4 if fw.domain == 0 else 8
    for x0 in (None, _t(np.tile(g["x0"], (n // 2048 + 1, 1))[:n])):
        wo, p = s.plugin_sample(wi, x0, T=T, variant=variant, seed=3, offset=11)
        pl = s.plugin_pdf(wi, wl, T=T, variant=variant)
        wo2, p2, pl2 = s.plugin_sample_pdf(wi, wl, x0, T=T, variant=variant, seed=3, offset=11)
        # (two different instruction streams of the same arithmetic: the compiler contracts a*b+c into an fma in one and not in
        #  the other here and there, and the flow amplifies the last bit — measured <= 4.3e-6; a flipped von Mises accept test or a
        #  wrong lane would be O(1))
        assert torch.allclose(wo, wo2, rtol=0, atol=1e-5)
        assert same_density(p2, p) and same_density(pl2, pl)
    with pytest.raises(RuntimeError, match="sample_pdf|null|wl"):
        from bsdf_diffusion_sampling_amd import _lib
        import ctypes as C
        _lib.check(_lib.lib().bsdfd_plugin_sample_pdf(s._h, variant, C.c_void_p(wi.data_ptr()), None, None, 0, 0, n, T,
                                                      C.c_void_p(wo.data_ptr()), C.c_void_p(p.data_ptr()), None, None))


def test_plugin_core_sample_pdf_t():
    from bsdf_diffusion_sampling_amd.brdf_measured_disk import MyBSDF
    plug = MyBSDF({"filename": "chm_orange_rgb", "measured": False})
    rng = np.random.default_rng(3)
    z, ph = rng.uniform(0.1, 1.0, size=5000), rng.uniform(0, 2 * np.pi, size=5000)
    wi = _t(np.stack([np.sqrt(1 - z * z) * np.cos(ph), np.sqrt(1 - z * z) * np.sin(ph), z], 1))
    wl = wi.flip(0).contiguous()
    wo, po, pl = plug.sample_pdf_t(wi, wl, seed=7)
    wo2, po2 = plug.sample_t(wi, seed=7)
    assert torch.allclose(wo, wo2, atol=2e-6, rtol=0) and same_density(po, po2)
    assert same_density(pl, plug.pdf_t(wi, wl))


@pytest.mark.parametrize("stem,variant", [("chm_orange_rgb_disk", 0), ("aniso_miro_7_rgb_disk", 0), ("aniso_miro_7_rgb_spherical", 0),
                                          ("bsdf_3_spherical", 1), ("aniso_miro_7_rgb_spherical_complex", 0)])
@pytest.mark.parametrize("n", [1, 17, 4097, 30001])
def test_per_query_context_gives_bit_identical_results(stem, variant, n):
    """bsdfd_plugin_sample_ex(opts.ctx_out) writes what depends on wi alone (conditioning term of layer 1, base-net outputs);
    bsdfd_plugin_pdf_ex(opts.ctx_in) reads it instead of re-evaluating the prologue the reference evaluates once per Euler step
    (rendering/utils/model.py:494) and twice per sample (mlp_brdf_sampling.py:20,24).  Results must not move by a bit,
    for any T, ragged sizes, injected and in-kernel base draws."""
    g, fw = load_case(stem)
    s = _sampler(fw, "split3")
    rng = np.random.default_rng(n)
    def dirs(lo):
        z, ph = rng.uniform(lo, 1.0, size=n), rng.uniform(0, 2 * np.pi, size=n)
        r = np.sqrt(1 - z * z)
        return _t(np.stack([r * np.cos(ph), r * np.sin(ph), z], 1))
    wi, wl = dirs(0.05), dirs(-1.0 if variant else 0.02)
    T = 4 if fw.domain == 0 else 8
    ctx = s.new_context(n)
    # (opaque; 144 B per query for the 32-wide nets in either tiling: per 16-query tile NM x 64 + 16 records of 16 B, per 32-query
    #  tile 4 x 64 + 32, plus one spare tile per segment)
    per_tile = ((fw.width // 16) * 64 + 16) * 16 if s.tile == 16 else (4 * 64 + 32) * 16
    assert ctx.numel() * 4 == ((n + s.tile - 1) // s.tile + 1) * per_tile
    ctx.fill_(float("nan"))
    for x0 in (None, _t(np.tile(g["x0"], (n // 2048 + 1, 1))[:n])):
        wo, p = s.plugin_sample(wi, x0, T=T, variant=variant, seed=3, offset=11)
        wo_c, p_c = s.plugin_sample(wi, x0, T=T, variant=variant, seed=3, offset=11, ctx_out=ctx)
        assert torch.equal(wo, wo_c) and torch.equal(p, p_c)      # writing the context does not change sample()
        for Tp in (T, 3):                                          # the context does not depend on T
            ref = s.plugin_pdf(wi, wl, T=Tp, variant=variant)
            assert torch.equal(ref, s.plugin_pdf(wi, wl, T=Tp, variant=variant, ctx_in=ctx))
            out = torch.empty_like(ref)
            assert s.plugin_pdf(wi, wl, T=Tp, variant=variant, ctx_in=ctx, out=out) is out and torch.equal(out, ref)
    # either call may write the context and either may read it (Mitsuba's path integrator asks eval_pdf() first, sample() second:
    # rendering/brdf_measured_disk.py:126,59): pdf writes -> sample and pdf read, bit-identical to the plain calls
    ctx2 = s.new_context(n)
    ctx2.fill_(float("nan"))
    ref_pdf = s.plugin_pdf(wi, wl, T=T, variant=variant)
    assert torch.equal(ref_pdf, s.plugin_pdf(wi, wl, T=T, variant=variant, ctx_out=ctx2))   # writing does not change pdf()
    same = lambda u, v: torch.equal(torch.nan_to_num(u, nan=-7.0), torch.nan_to_num(v, nan=-7.0))  # noqa: E731  (unused slots stay NaN)
    assert same(ctx2, ctx)                                                                    # ... and writes the same record
    wo_r, p_r = s.plugin_sample(wi, None, T=T, variant=variant, seed=3, offset=11)
    wo_c, p_c = s.plugin_sample(wi, None, T=T, variant=variant, seed=3, offset=11, ctx_in=ctx2)
    assert torch.equal(wo_r, wo_c) and torch.equal(p_r, p_c)
    assert torch.equal(ref_pdf, s.plugin_pdf(wi, wl, T=T, variant=variant, ctx_in=ctx2))
    with pytest.raises(RuntimeError, match="not both"):
        s.plugin_pdf(wi, wl, T=T, variant=variant, ctx_in=ctx, ctx_out=ctx2)
    with pytest.raises(RuntimeError, match="not both"):
        s.plugin_sample(wi, None, T=T, variant=variant, ctx_in=ctx, ctx_out=ctx2)
    # size and alignment are checked
    import ctypes as C
    from bsdf_diffusion_sampling_amd import _lib
    L = _lib.lib()
    p_ = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    with pytest.raises(RuntimeError, match="context"):
        s.plugin_pdf(wi, wl, T=T, variant=variant, ctx_in=ctx[:-64])
    with pytest.raises(RuntimeError, match="aligned"):
        o = _lib.Opts()
        o.ctx_in = ctx.data_ptr() + 4
        _lib.check(L.bsdfd_plugin_pdf_ex(s._h, variant, p_(wi), p_(wl), n, T, p_(p), C.byref(o), None))
    with pytest.raises(RuntimeError, match="not both"):   # through the raw C ABI as well
        _lib.check(L.bsdfd_plugin_pdf_ex(s._h, variant, p_(wi), p_(wl), n, T, p_(p), C.byref(_lib.opts(ctx_out=ctx, ctx_in=ctx2)), None))
    assert L.bsdfd_context_bytes(s._h, -1, 1) == -1 and L.bsdfd_context_bytes(s._h, 16, 0) == -1
    # NULL context = the plain calls
    _lib.check(L.bsdfd_plugin_pdf_ex(s._h, variant, p_(wi), p_(wl), n, T, p_(p), None, None))
    assert torch.equal(p, s.plugin_pdf(wi, wl, T=T, variant=variant))
    # rng_index: the Philox counter of row i is offset + rng_index[i] — a permuted batch draws what the original draws
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(n)).to(_dev())
    wo_a, p_a = s.plugin_sample(wi, None, T=T, variant=variant, seed=3, offset=11)
    wo_b, p_b = s.plugin_sample(wi[perm].contiguous(), None, T=T, variant=variant, seed=3, offset=11, rng_index=perm)
    assert torch.equal(wo_a[perm], wo_b) and torch.equal(p_a[perm], p_b)
    with pytest.raises(RuntimeError, match="rng_index"):
        s.plugin_sample(wi, None, T=T, variant=variant, rng_index=perm.int())


def test_plugin_core_context_cache():
    """MyBSDF.pdf(si, wl) / eval_pdf and MyBSDF.sample(si) for the same si, in EITHER order (Mitsuba's path integrator:
    eval_pdf first — rendering/brdf_measured_disk.py:126 -> :112 — then sample, :59): the core caches the per-query context
    keyed on the identity and version of si.wi; whichever call comes first fills it."""
    from bsdf_diffusion_sampling_amd.brdf_measured_spherical import MyBSDF
    from bsdf_diffusion_sampling_amd.plugin_base import SurfaceInteraction
    plug = MyBSDF({"filename": "chm_orange_rgb", "measured": False, "context_cache": True})   # opt-in since round 4
    off = MyBSDF({"filename": "chm_orange_rgb", "measured": False})
    assert off.context_cache is False
    rng = np.random.default_rng(3)
    z, ph = rng.uniform(0.1, 1.0, size=5000), rng.uniform(0, 2 * np.pi, size=5000)
    wi = _t(np.stack([np.sqrt(1 - z * z) * np.cos(ph), np.sqrt(1 - z * z) * np.sin(ph), z], 1))
    wl = wi.flip(0).contiguous()
    si = SurfaceInteraction(wi)
    hit = lambda p_, w: p_._ctx is not None and p_._ctx[0] == p_._wi_key(w)  # noqa: E731
    # sample -> pdf
    bs, _ = plug.sample(None, si, seed=5)
    bs_off, _ = off.sample(None, si, seed=5)
    assert hit(plug, wi) and off._ctx is None and torch.equal(bs.wo, bs_off.wo) and torch.equal(bs.pdf, bs_off.pdf)
    p_hit = plug.pdf(None, si, wl)
    assert torch.equal(p_hit, off.pdf(None, si, wl))
    # pdf / eval_pdf -> sample -> pdf: the pdf call fills, the other two read
    plug.invalidate_context()
    assert not hit(plug, wi)
    assert torch.equal(plug.pdf(None, si, wl), p_hit) and hit(plug, wi)
    fills = plug._ctx_unread_fills
    bs2, _ = plug.sample(None, si, seed=5)
    assert torch.equal(bs2.wo, bs.wo) and torch.equal(bs2.pdf, bs.pdf) and plug._ctx_unread_fills == 0 and fills == 1
    assert torch.equal(plug.pdf(None, si, wl), p_hit)
    # another wi tensor, or the same tensor modified in place: no hit, still correct
    wi2 = wi.clone()
    assert not hit(plug, wi2)
    wi.mul_(-1.0).mul_(-1.0)  # same values, version bumped
    assert not hit(plug, wi)
    assert torch.equal(plug.pdf(None, si, wl), p_hit)
    # a raw-pointer writer bumps the version explicitly
    plug.sample(None, si, seed=5)
    assert hit(plug, wi)
    torch.autograd.graph.increment_version(wi)
    assert not hit(plug, wi)
    plug.sample(None, si, seed=5)
    plug.invalidate_context()
    assert not hit(plug, wi)
    # a launch that raises leaves NO context behind (the entry used to be set before the launch)
    plug.sample(None, si, seed=5)
    assert hit(plug, wi)
    wi3 = wi.clone()
    with pytest.raises(RuntimeError):
        plug.sample(None, SurfaceInteraction(wi3), x0=torch.zeros(7, 2, device=wi.device))   # wrong x0 shape
    assert plug._ctx is None
    assert torch.equal(plug.pdf(None, SurfaceInteraction(wi3), wl), p_hit)
    # replacing the sampler invalidates the cache (the key carries the handle)
    plug.sample(None, si, seed=5)
    assert hit(plug, wi)
    old = plug.sampler
    plug.sampler = off.sampler
    assert not hit(plug, wi)
    plug.sampler = old
    # two streams: fill on one, read on the other, refill on the first — ordered by the event behind every launch
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for k in range(4):
        wik = (wi if k % 2 == 0 else wi2)
        with torch.cuda.stream(s1):
            a_ = plug.sample(None, SurfaceInteraction(wik), seed=5)[0]
        with torch.cuda.stream(s2):
            pk = plug.pdf(None, SurfaceInteraction(wik), wl)
        torch.cuda.synchronize()
        assert torch.equal(pk, p_hit) and torch.equal(a_.wo, bs.wo)
    # over the cap: no cache, same results
    small = MyBSDF({"filename": "chm_orange_rgb", "measured": False, "context_cache": True, "context_cache_max_bytes": 1024})
    small.sample(None, si, seed=5)
    assert small._ctx is None and torch.equal(small.pdf(None, si, wl), p_hit)
    # a host that never presents the same tensor twice: filling stops after `context_cache_patience` unread fills
    lone = MyBSDF({"filename": "chm_orange_rgb", "measured": False, "context_cache": True})
    for k in range(8):
        lone.sample(None, SurfaceInteraction(wi.clone()), seed=5)
    assert lone._ctx_unread_fills == lone.context_cache_patience and lone._ctx is None


PLUGIN_CASES = ["aniso_miro_7_rgb_disk", "chm_orange_rgb_disk", "vch_silk_blue_rgb_disk", "aniso_miro_7_rgb_spherical",
                "chm_orange_rgb_spherical", "bsdf_3_spherical"]


@pytest.mark.parametrize("stem", PLUGIN_CASES + ["chm_orange_rgb_spherical_n16k"])
def test_plugin_level_vs_reference_plugin_goldens(stem):
    """Plugin-level parity against tests/golden/<stem>_plugin.npz: the reference's own operators followed by the plugins'
    tensor ops (rendering/brdf_measured_disk.py:59-82,112-124, brdf_measured_spherical.py:35-39,69-91,122-137,
    bsdf_myresult.py:59-84,115-133), in fp32 as the plugins run them and in fp64.
    Bounds (north_star: 1e-4 on directions and pdfs, against the fp64 run): directions p99 <= 1e-5 and max <= 1e-4; pdf p99 <=
    1e-4 — FIXED bounds since round 4: the reference's own fp32 run is 1.4e-4 / 2.1e-4 away from its fp64 run on chm_orange
    (acos as written near the pole; recorded below as ref32_*), the kernel evaluates the same angle in a well-conditioned
    form (csrc/bsdfd.hip, cart_to_spher) and stays below 7e-5 / 1e-5 on every fixture."""
    from bsdf_diffusion_sampling_amd import _lib
    from conftest import GOLDEN
    import os
    _, fw = load_case(stem)
    p = np.load(os.path.join(GOLDEN, stem + "_plugin.npz"))
    s = _sampler(fw, "split3")
    T, full = int(p["meta_T"]), bool(p["meta_full_sphere"])
    variant = _lib.PLUGIN_FULLSPHERE if full else _lib.PLUGIN_MEASURED
    ctx = s.new_context(p["wi3"].shape[0])
    wo, pdf = s.plugin_sample(_t(p["wi3"]), _t(p["x0"]), T=T, variant=variant, ctx_out=ctx)
    wo, pdf = wo.cpu().numpy().astype(np.float64), pdf.cpu().numpy().astype(np.float64)
    ref_wo, ref_pdf = p["sample_wo3_f64"], p["sample_pdf_sa_f64"]
    # the fp32 reference's own distance from its fp64 run = the noise floor of this comparison
    ok = np.abs(ref_pdf) > 1e-6 * np.percentile(np.abs(ref_pdf), 99)
    noise_wo = np.abs(p["sample_wo3"] - ref_wo)
    noise_pdf = _rel(p["sample_pdf_sa"].astype(np.float64), ref_pdf)[ok]
    err_wo, err_pdf = np.abs(wo - ref_wo), _rel(pdf, ref_pdf)[ok]
    _record(f"plugin_sample[{stem}]", wo_p99=np.percentile(err_wo, 99), wo_max=err_wo.max(), pdf_median=np.median(err_pdf),
            pdf_p99=np.percentile(err_pdf, 99), ref32_wo_p99=np.percentile(noise_wo, 99), ref32_wo_max=noise_wo.max(),
            ref32_pdf_p99=np.percentile(noise_pdf, 99))
    assert np.percentile(err_wo, 99) <= 1e-5
    assert err_wo.max() <= 1e-4
    assert np.percentile(err_pdf, 99) <= 1e-4
    _tail_bound(f"plugin_sample_tail[{stem}]", err_pdf, noise_pdf)
    # ... and against the reference's OWN fp32 outputs (north_star: "match the reference path"): at most the two distances from
    # the fp64 run added up (triangle inequality on the percentiles, 25 % slack: percentiles are not norms).  On chm_orange
    # spherical this distance is ABOVE 1e-4 by design — the reference's fp32 acos(z / (r + 1e-8)) loses 6.5e-5 rad near the pole
    # (rendering/brdf_measured_spherical.py:35-39) and the kernel evaluates the same angle well-conditioned
    vs_ref32 = _rel(pdf, p["sample_pdf_sa"].astype(np.float64))[ok & (p["sample_pdf_sa"] != 0)]
    _record(f"plugin_sample_vs_ref32[{stem}]", vs_ref32_p99=np.percentile(vs_ref32, 99), kernel_vs_fp64_p99=np.percentile(err_pdf, 99),
            ref32_vs_fp64_p99=np.percentile(noise_pdf, 99))
    assert np.percentile(vs_ref32, 99) <= 1.25 * (np.percentile(err_pdf, 99) + np.percentile(noise_pdf, 99))
    # guards: rows the reference zeroes are zero here (rows that sit within fp32 noise of a threshold excepted)
    z_ref, z_got = p["sample_pdf_sa"] == 0, pdf == 0
    decided = (ref_pdf == 0) | (np.abs(ref_pdf) > 1e-30)   # (a density that underflows fp32 may be 0 or a subnormal on either side)
    assert ((z_ref != z_got) & decided).sum() <= 2
    if fw.domain == 0:
        bad = (p["sample_wo3"][:, 0] == 0) & (p["sample_wo3"][:, 1] == 0)
        assert bad.sum() >= 1 and np.all(pdf[bad] == 0) and np.all(wo[bad] == np.array([0.0, 0.0, 1.0]))
    # pdf(): the reference's fp32 run is the only run there is (mlp_brdf_sampling.py:71,146): compare with the fp64 oracle
    # (pinned to the reference's plugin level by tests/test_oracle_golden.py) and with the fp32 fixture
    orc = O.Oracle(fw)
    for wi3, wo3, key, use_ctx in ((p["pdf_wi3"], p["pdf_wo3"], "pdf_sa", False), (p["wi3"], p["sample_wo3"], "pdf_sa_of_samples", True)):
        got = s.plugin_pdf(_t(wi3), _t(wo3), T=T, variant=variant, ctx_in=ctx if use_ctx else None).cpu().numpy().astype(np.float64)
        if fw.domain == 0:
            want = O.plugin_pdf_disk(orc, wi3, wo3, T=T)
        else:
            want = O.plugin_pdf_spherical(orc, wi3.astype(np.float64), wo3.astype(np.float64), T=T, full_sphere=full)
        okp = np.abs(want) > 1e-6 * np.percentile(np.abs(want), 99)
        e, n32 = _rel(got, want)[okp], _rel(p[key].astype(np.float64), want)[okp]
        vs32 = _rel(got, p[key].astype(np.float64))[okp & (p[key] != 0)]
        _record(f"plugin_pdf[{stem}:{key}]", pdf_median=np.median(e), pdf_p99=np.percentile(e, 99), ref32_pdf_p99=np.percentile(n32, 99),
                vs_ref32_p99=np.percentile(vs32, 99))
        assert np.percentile(e, 99) <= 1e-4
        _tail_bound(f"plugin_pdf_tail[{stem}:{key}]", e, n32)
        assert np.percentile(vs32, 99) <= 1.25 * (np.percentile(e, 99) + np.percentile(n32, 99))   # vs the reference's own fp32 run
        assert (((p[key] == 0) != (got == 0)) & ((want == 0) | (np.abs(want) > 1e-30))).sum() <= 2
    if not full:
        got = s.plugin_pdf(_t(p["pdf_wi3"]), _t(p["pdf_wo3"]), T=T, variant=variant).cpu().numpy()
        assert np.all(got[:16] == 0)  # cos(theta_o) <= 0 and cos(theta_i) <= 0 lanes


@pytest.mark.parametrize("stem,variant", [("chm_orange_rgb_disk", 0), ("aniso_miro_7_rgb_spherical", 0)])
def test_fused_sample_pdf_in_f16_precision(stem, variant):
    """The fused sample+pdf kernels also exist in the single-pass fp16 mode (tcnn class, 1e-2): same directions and
    densities as the two single-op calls of that mode, and close to the split3 results."""
    g, fw = load_case(stem)
    s16, s3 = _sampler(fw, "f16"), _sampler(fw, "split3")
    rng = np.random.default_rng(5)
    n = 20000
    def dirs(lo):
        z, ph = rng.uniform(lo, 1.0, size=n), rng.uniform(0, 2 * np.pi, size=n)
        r = np.sqrt(1 - z * z)
        return _t(np.stack([r * np.cos(ph), r * np.sin(ph), z], 1))
    wi, wl = dirs(0.05), dirs(0.02)
    x0 = _t(np.tile(g["x0"], (n // 2048 + 1, 1))[:n])
    T = 4 if fw.domain == 0 else 8
    wo, p = s16.plugin_sample(wi, x0, T=T, variant=variant)
    pl = s16.plugin_pdf(wi, wl, T=T, variant=variant)
    wo2, p2, pl2 = s16.plugin_sample_pdf(wi, wl, x0, T=T, variant=variant)
    assert torch.allclose(wo, wo2, atol=2e-3, rtol=0)
    # plain-fp16 contractions carry 2^-11 operand error, which the flow amplifies into 3e-3 .. 9e-2 on the density:
    # two evaluation orders of that mode (forward-mode tangents in the spherical fused kernel, meet-in-the-middle elsewhere)
    # agree at that level, not at fp32 noise
    for a_, b_ in ((p2, p), (pl2, pl)):
        a_, b_ = a_.cpu().numpy().astype(np.float64), b_.cpu().numpy().astype(np.float64)
        ok = np.abs(b_) > 1e-6 * np.percentile(np.abs(b_), 99)
        rel = np.abs(a_ - b_)[ok] / np.abs(b_[ok])
        assert np.isfinite(a_).all() and np.median(rel) < 2e-2 and np.percentile(rel, 99) < 0.5, (np.median(rel), np.percentile(rel, 99))
    wo3, _ = s3.plugin_sample(wi, x0, T=T, variant=variant)
    # (a row within fp16 noise of the r^2 >= 0.995 / cos guards may flip to the guarded value in one precision only)
    assert ((wo2 - wo3).abs().max(dim=1).values > 2e-2).float().mean().item() < 2e-3
