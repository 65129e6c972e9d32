"""GPU tests at BASELINE.json's full sizes through size-independent properties, plus the
mixed-material table (configs[1..3]).  The oracle checks a random subsample (it would
need minutes on the whole batch); the rest are invariants of the domain:
  * split invariance: running a batch in two halves (Philox offset = global index) is
    bit-identical to running it whole — covers the tile->wave map, ragged tails and the
    sharding rule (a GPU's shard is such a slice);
  * determinism (same seed -> same bits), seed sensitivity;
  * produced directions are unit vectors, guarded rows have pdf == 0, nothing is NaN;
  * sample() -> pdf() agree in the bulk only as far as the reference itself does
    (forward alpha=t/T vs reverse 1-t/T): median within 20 %.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from conftest import same_density,  load_case  # noqa: E402
from oracle import bsdf_oracle as O  # noqa: E402


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def _wi(domain, n, seed):
    import bench
    return bench.make_wi(domain, n, seed, _dev())


@pytest.mark.parametrize("stem,domain,n,T", [("aniso_miro_7_rgb_disk", "disk", 1 << 20, 8),
                                             ("aniso_miro_7_rgb_spherical", "spherical", 1 << 24, 8)])
def test_full_size_properties(stem, domain, n, T):
    from bsdf_diffusion_sampling_amd.sampler import FlowSampler
    g, fw = load_case(stem)
    s = FlowSampler(fw)
    wi = _wi(domain, n, 1234)
    wo, pdf = s.plugin_sample(wi, None, T=T, seed=99)
    wo2, pdf2 = s.plugin_sample(wi, None, T=T, seed=99)
    assert torch.equal(wo, wo2) and torch.equal(pdf, pdf2)
    wo3, _ = s.plugin_sample(wi, None, T=T, seed=100)
    assert not torch.equal(wo, wo3)
    # split invariance with a ragged cut
    cut = n // 2 + 7
    a_wo, a_pdf = s.plugin_sample(wi[:cut].contiguous(), None, T=T, seed=99, offset=0)
    b_wo, b_pdf = s.plugin_sample(wi[cut:].contiguous(), None, T=T, seed=99, offset=cut)
    assert torch.equal(torch.cat([a_wo, b_wo]), wo) and torch.equal(torch.cat([a_pdf, b_pdf]), pdf)
    assert torch.isfinite(wo).all() and torch.isfinite(pdf).all()
    assert torch.allclose((wo * wo).sum(1), torch.ones(n, device=_dev()), atol=2e-5)
    if domain == "disk":
        guarded = (wo[:, 0] == 0) & (wo[:, 1] == 0)
        assert torch.all(pdf[guarded] == 0) and torch.all(wo[:, 2] >= 0)
    else:
        assert torch.all(pdf[wo[:, 2] <= 0] == 0)
    # pdf() of the produced directions: finite, same split invariance, consistent in the bulk
    p = s.plugin_pdf(wi, wo, T=T)
    pa = s.plugin_pdf(wi[:cut].contiguous(), wo[:cut].contiguous(), T=T)
    assert torch.equal(pa, p[:cut]) and torch.isfinite(p).all()
    ok = (pdf > 1e-3) & (p > 1e-3)
    ratio = (p[ok] / pdf[ok]).cpu().numpy()
    assert 0.8 < np.median(ratio) < 1.25
    # per-query context at full size: sample() writes it, pdf() of the same wavefront reads it — not a bit moves
    ctx = s.new_context(n)
    wo_c, pdf_c = s.plugin_sample(wi, None, T=T, seed=99, ctx_out=ctx)
    assert torch.equal(wo_c, wo) and torch.equal(pdf_c, pdf)
    assert torch.equal(s.plugin_pdf(wi, wo, T=T, ctx_in=ctx), p)
    del ctx, wo_c, pdf_c
    # oracle on a random subsample, with the base draw injected so both sides flow the same x0
    idx = torch.randperm(n, generator=torch.Generator().manual_seed(5))[:4096]
    wis = wi[idx.to(_dev())].contiguous()
    orc = O.Oracle(fw)
    wi_np = wis.cpu().numpy().astype(np.float64)
    cond = wi_np[:, :2] if domain == "disk" else O.cart_to_spher(wi_np)
    rng = np.random.default_rng(7)
    if domain == "disk":
        x0 = orc.base_sample(cond, rng.standard_normal((4096, 2)))
        wo_o, pdf_o = O.plugin_sample_disk(orc, wi_np, x0, T=T)
    else:
        mu, kappa = orc.base_von_mises_params(cond)
        x0 = orc.base_sample(cond, rng.standard_normal(4096), phi=rng.vonmises(mu, kappa))
        wo_o, pdf_o = O.plugin_sample_spherical(orc, wi_np, x0, T=T)
    # both sides flow the SAME fp32 base draw (the oracle in fp64 arithmetic, and once more in fp32 arithmetic: the
    # distance between those two is the reference-class fp32 noise of these very rows, incl. the plugin's fp32
    # acos / atan2 / sincos chain — rendering/brdf_measured_spherical.py:35-39)
    x0 = x0.astype(np.float32)
    x0_t = torch.from_numpy(x0).to(_dev())
    wi32 = wis.cpu().numpy()
    orc32 = O.Oracle(fw, np.float32)
    if domain == "disk":
        wo_o, pdf_o = O.plugin_sample_disk(orc, wi_np, x0.astype(np.float64), T=T)
        wo_32, pdf_32 = O.plugin_sample_disk(orc32, wi32, x0, T=T)
    else:
        wo_o, pdf_o = O.plugin_sample_spherical(orc, wi_np, x0.astype(np.float64), T=T)
        wo_32, pdf_32 = O.plugin_sample_spherical(orc32, wi32, x0, T=T)
    wo_s, pdf_s = s.plugin_sample(wis, x0_t, T=T)
    _, acc = orc.flow(x0.astype(np.float64), cond, T, reverse=False)
    sel = (np.abs(acc) > 1e-3) & (np.abs(acc) < 1e3) & (np.abs(pdf_o) > 1e-6 * np.percentile(np.abs(pdf_o), 99))
    err_wo, noise_wo = np.abs(wo_s.cpu().numpy() - wo_o), np.abs(wo_32.astype(np.float64) - wo_o)
    rel = np.abs(pdf_s.cpu().numpy() - pdf_o)[sel] / np.abs(pdf_o[sel])
    noise = np.abs(pdf_32.astype(np.float64) - pdf_o)[sel] / np.abs(pdf_o[sel])
    from test_gpu_parity import _record
    _record(f"full_size_subsample[{stem}:{n}]", wo_p99=np.percentile(err_wo, 99), wo_max=err_wo.max(), pdf_median=np.median(rel),
            pdf_p99=np.percentile(rel, 99), fp32_oracle_wo_max=noise_wo.max(), fp32_oracle_pdf_p99=np.percentile(noise, 99))
    assert np.percentile(err_wo, 99) <= 1e-5 and err_wo.max() <= 1e-4
    assert np.percentile(rel, 99) <= 1e-4, (np.percentile(rel, 99), np.percentile(noise, 99))
    from test_gpu_parity import _tail_bound
    _tail_bound(f"full_size_tail[{stem}:{n}]", rel, noise)


def test_mixed_material_table_matches_per_material_calls():
    from bsdf_diffusion_sampling_amd.materials import MaterialTable
    stems = ["aniso_miro_7_rgb_disk", "chm_orange_rgb_disk", "vch_silk_blue_rgb_disk", "aniso_sari_silk_2color_rgb_disk"]
    tab = MaterialTable(stems)
    n = 50000
    wi = _wi("disk", n, 3)
    ids = torch.randint(0, len(stems), (n,), generator=torch.Generator().manual_seed(1)).to(_dev())
    ids[:100] = 3  # make sure ordering inside a bucket is exercised
    g, _ = load_case("chm_orange_rgb_disk")
    x0 = torch.from_numpy(np.tile(g["x0"], (n // 2048 + 1, 1))[:n]).to(_dev())
    wo, pdf = tab.sample(ids, wi, x0=x0)                       # ONE segmented launch for the 4 disk materials
    p = tab.pdf(ids, wi, wo)
    for m in range(len(stems)):
        sel = (ids == m).nonzero()[:, 0]
        wo_m, pdf_m = tab.samplers[m].plugin_sample(wi[sel].contiguous(), x0[sel].contiguous(), T=4)
        assert torch.equal(wo[sel], wo_m) and torch.equal(pdf[sel], pdf_m)
        assert torch.equal(p[sel], tab.samplers[m].plugin_pdf(wi[sel].contiguous(), wo[sel].contiguous(), T=4))
    # segmented == one launch per bucket, bit for bit, also with the in-kernel RNG (counter = offset + lane index)
    a = tab.sample(ids, wi, seed=11, offset=5)
    b = tab.sample(ids, wi, seed=11, offset=5, segmented=False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(tab.pdf(ids, wi, a[0]), tab.pdf(ids, wi, a[0], segmented=False))
    # sample(wi) + pdf(wi, wl) of the same intersections in one launch per kernel signature
    wl = _wi("disk", n, 9)
    wo_f, po_f, pl_f = tab.sample_pdf(ids, wi, wl, seed=11, offset=5)
    # (the fused kernel carries the Jacobian in forward mode, the single-op kernels form it by meeting in the middle: the same
    #  numbers up to fp32 noise, conftest.same_density)
    assert torch.allclose(wo_f, a[0], atol=2e-6, rtol=0) and same_density(po_f, a[1])
    assert same_density(pl_f, tab.pdf(ids, wi, wl))
    # a bucketing plan computed once serves both calls
    plan = tab.bucket(ids)
    c = tab.sample(plan, wi, seed=11, offset=5)
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1]) and torch.equal(tab.pdf(plan, wi, a[0]), tab.pdf(ids, wi, a[0]))
    # per-query contexts: sample(ctx=) fills them, pdf(ctx=) of the same plan and wi reads them — bit-identical
    cx = {}
    d = tab.sample(plan, wi, seed=11, offset=5, ctx=cx)
    assert torch.equal(a[0], d[0]) and torch.equal(a[1], d[1]) and len(cx) == 1
    assert torch.equal(tab.pdf(plan, wi, wl, ctx=cx), tab.pdf(plan, wi, wl))
    with pytest.raises(ValueError, match="context"):
        tab.pdf(plan, wi, wl, ctx={})
    # the other order (a path tracer asks pdf for the emitter sample first): pdf fills, sample reads
    cy = {}
    assert torch.equal(tab.pdf(plan, wi, wl, ctx=cy, ctx_fill=True), tab.pdf(plan, wi, wl))
    e = tab.sample(plan, wi, seed=11, offset=5, ctx=cy, ctx_fill=False)
    assert torch.equal(a[0], e[0]) and torch.equal(a[1], e[1])
    # the dict keeps one buffer per (kernel signature, run): wavefronts with other bucket sizes reuse it, a reader with another
    # layout than the last fill is refused, and the per-bucket path refuses ctx= instead of ignoring it
    ids_b = torch.randint(0, len(tab), (n // 2,), generator=torch.Generator().manual_seed(44)).to(_dev())
    for k in range(3):
        tab.sample(ids_b if k % 2 else ids, wi[: n // 2] if k % 2 else wi, seed=11, ctx=cx)
    assert len(cx) == 1
    with pytest.raises(ValueError, match="context"):
        tab.pdf(ids_b, wi[: n // 2], wl[: n // 2], ctx=cx)       # cx was last filled for the full wavefront
    with pytest.raises(ValueError, match="segmented"):
        tab.sample(ids, wi, seed=11, ctx={}, segmented=False)
    # an id with no queries is fine; so is a single-row bucket
    ids2 = torch.zeros(1000, dtype=torch.int64, device=_dev())
    ids2[7] = 2
    wo2, pdf2 = tab.sample(ids2, wi[:1000].contiguous(), seed=3)
    wo3, pdf3 = tab.sample(ids2, wi[:1000].contiguous(), seed=3, segmented=False)
    assert torch.isfinite(wo2).all() and torch.equal(wo2, wo3) and torch.equal(pdf2, pdf3)


def test_plan_with_extra_bins_serves_sample_and_pdf():
    """A bucketing plan with `extra_bins` (a renderer's floor hits / misses: lanes that carry no material) may be
    passed to sample(), pdf() and sample_pdf(): the material lanes equal a table call on those lanes alone, the
    other lanes come back as ZEROS (never uninitialised memory), segmented or not; malformed plans are rejected."""
    from bsdf_diffusion_sampling_amd.materials import MaterialTable
    stems = ["aniso_miro_7_rgb_disk", "chm_orange_rgb_disk", "vch_silk_blue_rgb_disk"]
    tab = MaterialTable(stems)
    n = 20000
    wi = _wi("disk", n, 5)
    ids = torch.randint(0, len(stems) + 2, (n,), generator=torch.Generator().manual_seed(2)).to(_dev())
    plan = tab.bucket(ids, extra_bins=2)
    mat = ids < len(stems)
    assert 0 < int(mat.sum()) < n
    for seg in (True, False):
        wo, pdf = tab.sample(plan, wi, seed=4, offset=7, segmented=seg, rng="bucketed")
        assert torch.count_nonzero(wo[~mat]) == 0 and torch.count_nonzero(pdf[~mat]) == 0
        assert torch.isfinite(wo).all() and torch.isfinite(pdf).all()
        assert torch.allclose((wo[mat] ** 2).sum(1), torch.ones(int(mat.sum()), device=_dev()), atol=1e-4)
        # rng="bucketed": the material lanes keep their relative (bucketed) order when the other lanes are dropped, so a
        # table call on them alone sees the same Philox counters
        wo_m, pdf_m = tab.sample(ids[mat], wi[mat].contiguous(), seed=4, offset=7, segmented=seg, rng="bucketed")
        assert torch.equal(wo[mat], wo_m) and torch.equal(pdf[mat], pdf_m)
        # rng="lane" (the default): the counter of a lane is offset + its index in the callers' array, whatever the
        # other lanes carry — relabelling the non-material lanes as one more material does not move the material lanes
        wo_l, pdf_l = tab.sample(plan, wi, seed=4, offset=7, segmented=seg)
        ids_all = torch.where(mat, ids, torch.zeros_like(ids))
        wo_a, pdf_a = tab.sample(ids_all, wi, seed=4, offset=7, segmented=seg)
        assert torch.equal(wo_l[mat], wo_a[mat]) and torch.equal(pdf_l[mat], pdf_a[mat])
        assert not torch.equal(wo_l[mat], wo[mat])
        p = tab.pdf(plan, wi, wo, segmented=seg)
        assert torch.count_nonzero(p[~mat]) == 0
        assert torch.equal(p[mat], tab.pdf(ids[mat], wi[mat].contiguous(), wo[mat].contiguous(), segmented=seg))
    # bucketed flow (gather once, keep bucket order through sample() and pdf(), scatter once) == the per-call flow
    wi_b = tab.gather(plan, wi)
    wo_b, pdf_b = tab.sample(plan, wi_b, seed=4, offset=7, bucketed=True)
    p_b = tab.pdf(plan, wi_b, wo_b, bucketed=True)
    wo_r, pdf_r, p_r = tab.scatter(plan, wo_b, pdf_b, p_b)
    wo_c, pdf_c = tab.sample(plan, wi, seed=4, offset=7)
    assert torch.equal(wo_r, wo_c) and torch.equal(pdf_r, pdf_c) and torch.equal(p_r, tab.pdf(plan, wi, wo_c))
    with pytest.raises(ValueError):
        tab.sample(plan, wi, bucketed=True)                           # caller-order array passed as bucketed
    with pytest.raises(ValueError):
        tab.sample(ids, wi_b, bucketed=True)                          # bucketed needs a plan
    wl = _wi("disk", n, 6)
    wo_f, po_f, pl_f = tab.sample_pdf(plan, wi, wl, seed=4, offset=7)
    assert torch.count_nonzero(wo_f[~mat]) == 0 and torch.count_nonzero(pl_f[~mat]) == 0
    with pytest.raises(ValueError):
        tab.sample((plan[0][:-1], plan[1]), wi)                       # perm and counts disagree
    with pytest.raises(ValueError):
        tab.sample(plan, wi[:100].contiguous())                       # plan of another batch size
    with pytest.raises(ValueError):
        tab.pdf((plan[0], plan[1][:2]), wi, wo)                       # fewer bins than materials
    with pytest.raises(RuntimeError):
        tab.sample(plan, wi.double())                                 # inputs are validated before pointer arithmetic
    with pytest.raises(RuntimeError):
        tab.pdf(plan, wi, wo.t().contiguous().t())                    # non-contiguous


def test_mixed_domains_and_more_than_64_buckets():
    """Disk + spherical + full-sphere materials interleaved by id (runs of adjacent buckets per kernel
    signature), and a table of 77 materials (> 64 buckets: the ABI splits the launch)."""
    from bsdf_diffusion_sampling_amd import weights as W
    from bsdf_diffusion_sampling_amd.materials import MaterialTable
    stems = ["chm_orange_rgb_disk", "chm_orange_rgb_spherical", "bsdf_3_spherical", "aniso_miro_7_rgb_disk",
             "aniso_miro_7_rgb_spherical", "bsdf_7_spherical", "vch_silk_blue_rgb_disk"]
    tab = MaterialTable(stems)
    n = 30011
    wi = _wi("spherical", n, 9)
    ids = torch.randint(0, len(stems), (n,), generator=torch.Generator().manual_seed(4)).to(_dev())
    a = tab.sample(ids, wi, seed=2)
    b = tab.sample(ids, wi, seed=2, segmented=False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(tab.pdf(ids, wi, a[0]), tab.pdf(ids, wi, a[0], segmented=False))
    big = MaterialTable(W.list_shipped("disk") + W.list_shipped("spherical"))
    assert len(big) == 77
    ids = torch.randint(0, 77, (n,), generator=torch.Generator().manual_seed(6)).to(_dev())
    a = big.sample(ids, wi, seed=8)
    b = big.sample(ids, wi, seed=8, segmented=False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.isfinite(a[1]).all()
    # per-query contexts across interleaved runs and a launch split at 64 buckets (context slots are numbered by bucket)
    for t in (tab, big):
        idt = ids % len(t)
        cx = {}
        wo_c, pdf_c = t.sample(idt, wi, seed=8, ctx=cx)
        wo_r, pdf_r = t.sample(idt, wi, seed=8)
        assert torch.equal(wo_c, wo_r) and torch.equal(pdf_c, pdf_r) and len(cx) >= 2
        wl = _wi("spherical", n, 21)
        assert torch.equal(t.pdf(idt, wi, wl, ctx=cx), t.pdf(idt, wi, wl))


@pytest.mark.parametrize("tile", [16, 32])
def test_buckets_smaller_than_a_tile(tile, monkeypatch):
    """Segmented launches whose buckets are smaller than one wave tile (1 .. 40 queries per material, some empty): every
    bucket starts its own partial tile with its own weight image; results equal the per-material calls bit for bit, for
    sample, pdf, the fused sample+pdf call and with a per-query context, in both tilings."""
    monkeypatch.setenv("BSDFD_TILE", str(tile))
    from bsdf_diffusion_sampling_amd.materials import MaterialTable
    stems = ["chm_orange_rgb_disk", "aniso_miro_7_rgb_disk", "vch_silk_blue_rgb_disk", "chm_orange_rgb_spherical",
             "aniso_miro_7_rgb_spherical", "bsdf_3_spherical", "bsdf_7_spherical"]
    tab = MaterialTable(stems)
    assert all(s.tile == tile for s in tab.samplers)
    for n, seed in ((7, 1), (70, 2), (131, 3)):
        wi, wl = _wi("spherical", n, 40 + seed), _wi("spherical", n, 50 + seed)
        ids = torch.randint(0, len(stems), (n,), generator=torch.Generator().manual_seed(seed)).to(_dev())
        ids[ids == 2] = 3                       # an empty bucket in the middle
        a = tab.sample(ids, wi, seed=5, offset=3)
        b = tab.sample(ids, wi, seed=5, offset=3, segmented=False)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.isfinite(a[1]).all()
        pa = tab.pdf(ids, wi, wl)
        assert torch.equal(pa, tab.pdf(ids, wi, wl, segmented=False))
        for m in range(len(stems)):
            sel = (ids == m).nonzero()[:, 0]
            if len(sel):
                T = 4 if stems[m].endswith("_disk") else 8
                variant = 1 if stems[m].startswith("bsdf_") else 0
                assert torch.equal(pa[sel], tab.samplers[m].plugin_pdf(wi[sel].contiguous(), wl[sel].contiguous(), T=T, variant=variant))
        cx = {}
        wo_c, pdf_c = tab.sample(ids, wi, seed=5, offset=3, ctx=cx)
        assert torch.equal(wo_c, a[0]) and torch.equal(pdf_c, a[1])
        assert torch.equal(tab.pdf(ids, wi, wl, ctx=cx), pa)
        wo_f, po_f, pl_f = tab.sample_pdf(ids, wi, wl, seed=5, offset=3)
        from conftest import same_density
        assert torch.allclose(wo_f, a[0], atol=2e-5) and same_density(po_f, a[1]) and same_density(pl_f, pa)


def test_all_shipped_weight_sets_run_and_are_sane():
    """Every shipped (material, domain) handle builds and samples (config 4's 52 measured sets + 25 bsdf)."""
    from bsdf_diffusion_sampling_amd import _lib, weights as W
    from bsdf_diffusion_sampling_amd.sampler import FlowSampler
    n = 4096
    for dom in ("disk", "spherical"):
        wi = _wi(dom, n, 11)
        for stem in W.list_shipped(dom):
            fw = W.load(W.shipped_path(stem[: -len(dom) - 1], dom))
            s = FlowSampler(fw)
            variant = _lib.PLUGIN_FULLSPHERE if stem.startswith("bsdf_") else _lib.PLUGIN_MEASURED
            wo, pdf = s.plugin_sample(wi, None, T=4 if dom == "disk" else 8, variant=variant, seed=1)
            assert torch.isfinite(wo).all() and torch.isfinite(pdf).all(), stem
            assert (pdf > 0).float().mean() > 0.5, stem
            s.close()


def test_every_shipped_material_matches_the_oracle():
    """Accuracy sweep over all 77 plugin weight sets (27 disk, 25 spherical, 25 full-sphere bsdf_<i>):
    flow + pdf vs the fp64 oracle on 1024 queries each, same tolerances as the golden cases."""
    from bsdf_diffusion_sampling_amd import weights as W
    from bsdf_diffusion_sampling_amd.sampler import FlowSampler
    rng = np.random.default_rng(2024)
    n = 1024
    worst = {}
    tail_worst = (0.0, "", 0.0)
    orc32_of = lambda fw_: O.Oracle(fw_, np.float32)  # noqa: E731
    for dom in ("disk", "spherical"):
        T = 4 if dom == "disk" else 8
        for stem in W.list_shipped(dom):
            fw = W.load(W.shipped_path(stem[: -len(dom) - 1], dom))
            orc = O.Oracle(fw)
            if dom == "disk":
                r, a = 0.95 * np.sqrt(rng.random(n)), 2 * np.pi * rng.random(n)
                cond = np.stack([r * np.cos(a), r * np.sin(a)], 1)
                x0 = orc.base_sample(cond, rng.standard_normal((n, 2)))
            else:
                hi = 3.0 if stem.startswith("bsdf_") else 1.5
                cond = np.stack([hi * rng.random(n), (2 * rng.random(n) - 1) * np.pi], 1)
                mu, kappa = orc.base_von_mises_params(cond)
                x0 = orc.base_sample(cond, rng.standard_normal(n), phi=rng.vonmises(mu, kappa))
            cond32, x032 = cond.astype(np.float32), x0.astype(np.float32)
            s = FlowSampler(fw)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(_dev())
            x, p = s.network_sampling(t(cond32), t(x032), T=T)
            x, p = x.cpu().numpy().astype(np.float64), p.cpu().numpy().astype(np.float64)
            xo, po = orc.network_sampling(cond32, x032, T)
            _, acc = orc.flow(x032, cond32, T, reverse=False)
            assert np.isfinite(p).all(), stem
            ok = (np.abs(acc) > 1e-3) & (np.abs(acc) < 1e3)
            ok &= np.abs(po) > 1e-6 * np.percentile(np.abs(po[ok]), 99)
            rel = np.abs(p - po)[ok] / np.abs(po[ok])
            xerr = np.abs(x - xo)[ok].max()
            worst[stem] = (np.percentile(rel, 99), xerr)
            assert xerr < 1e-4, (stem, xerr)
            assert np.percentile(rel, 99) < 1e-4, (stem, np.percentile(rel, 99))
            # tail (tests/test_gpu_parity.py::_tail_bound): the maximum against 4 x an fp32-arithmetic evaluation of the same rows
            _, p32 = orc32_of(fw).network_sampling(cond32, x032, T)
            rel32 = np.abs(p32.astype(np.float64) - po)[ok] / np.abs(po[ok])
            tail_worst = max(tail_worst, (float(rel.max()), stem, float(rel32.max())))
            assert rel.max() <= max(2e-3, 4.0 * rel32.max()), (stem, rel.max(), rel32.max())
            # reverse direction on the produced points
            pp = s.network_pdf(t(x.astype(np.float32)), t(cond32), T=T).cpu().numpy().astype(np.float64)
            ppo = orc.network_pdf(x.astype(np.float32), cond32, T)
            _, accr = orc.flow(x.astype(np.float32), cond32, T, reverse=True)
            okr = (np.abs(accr) > 1e-3) & (np.abs(accr) < 1e3)
            okr &= np.abs(ppo) > 1e-6 * np.percentile(np.abs(ppo[okr]), 99)
            relr = np.abs(pp - ppo)[okr] / np.abs(ppo[okr])
            assert np.percentile(relr, 99) < 1e-4, stem
            pp32 = orc32_of(fw).network_pdf(x.astype(np.float32), cond32, T)
            relr32 = np.abs(pp32.astype(np.float64) - ppo)[okr] / np.abs(ppo[okr])
            tail_worst = max(tail_worst, (float(relr.max()), stem + ":pdf", float(relr32.max())))
            assert relr.max() <= max(2e-3, 4.0 * relr32.max()), (stem, relr.max(), relr32.max())
            s.close()
    w = max(worst.items(), key=lambda kv: kv[1][0])
    print("worst p99 pdf rel-err:", w)
    from test_gpu_parity import _record
    _record("every_shipped_material", worst_p99=float(w[1][0]), worst_p99_set=w[0], worst_max=tail_worst[0], worst_max_set=tail_worst[1],
            fp32_reference_max_there=tail_worst[2])


@pytest.mark.parametrize("n,m", [(0, 3), (1, 1), (4095, 5), (4096, 64), (4097, 52), (1 << 20, 52), (3_000_001, 7)])
def test_native_bucketing_equals_stable_argsort(n, m):
    """csrc/bucket.hip: the stable counting sort reproduces torch.argsort(stable=True) + bincount exactly."""
    from bsdf_diffusion_sampling_amd.sharding import bucket_by_material
    ids = torch.randint(0, m, (n,), generator=torch.Generator().manual_seed(n + m))
    if n > 10:
        ids[: n // 3] = m - 1                       # a long run and an empty-ish tail of bins
    perm, counts = bucket_by_material(ids.to(_dev()), m)
    assert perm.dtype == torch.int64 and counts.dtype == torch.int64
    assert torch.equal(counts.cpu(), torch.bincount(ids, minlength=m))
    assert torch.equal(perm.cpu(), torch.argsort(ids, stable=True))


def test_bucketing_rejects_bad_ids_and_sizes():
    from bsdf_diffusion_sampling_amd.materials import MaterialTable
    from bsdf_diffusion_sampling_amd.sharding import bucket_by_material
    tab = MaterialTable(["aniso_miro_7_rgb_disk", "chm_orange_rgb_disk"])
    ids = torch.tensor([0, 1, 2, 0], device=_dev())
    with pytest.raises(ValueError, match="material ids"):
        tab.bucket(ids)
    # more than 64 materials: torch path, same contract
    big = torch.randint(0, 100, (5000,), generator=torch.Generator().manual_seed(0)).to(_dev())
    perm, counts = bucket_by_material(big, 100)
    assert torch.equal(perm.cpu(), torch.argsort(big.cpu(), stable=True)) and int(counts.sum()) == 5000


def test_128Mi_queries_in_one_launch():
    """BASELINE.json config 4 totals 128 Mi queries; one launch of that size must index correctly (64-bit row
    arithmetic): the head and the tail of the batch equal small launches with the matching Philox offsets."""
    from bsdf_diffusion_sampling_amd import weights as W
    from bsdf_diffusion_sampling_amd.sampler import FlowSampler
    s = FlowSampler(W.load(W.shipped_path("chm_orange_rgb", "disk")))
    n = (1 << 27) + 5
    base = _wi("disk", 1 << 20, 4)
    wi = base.repeat((n >> 20) + 1, 1)[:n].contiguous()
    wo = torch.empty_like(wi)
    pdf = torch.empty(n, device=_dev())
    s.plugin_sample(wi, None, T=1, seed=5, offset=0, out=(wo, pdf))
    torch.cuda.synchronize()
    k = 70000
    head = s.plugin_sample(wi[:k].contiguous(), None, T=1, seed=5, offset=0)
    tail = s.plugin_sample(wi[n - k:].contiguous(), None, T=1, seed=5, offset=n - k)
    assert torch.equal(wo[:k], head[0]) and torch.equal(pdf[:k], head[1])
    assert torch.equal(wo[n - k:], tail[0]) and torch.equal(pdf[n - k:], tail[1])
    mid = 3 * (1 << 25) + 17
    m = s.plugin_sample(wi[mid:mid + k].contiguous(), None, T=1, seed=5, offset=mid)
    assert torch.equal(wo[mid:mid + k], m[0]) and torch.equal(pdf[mid:mid + k], m[1])
    p = s.plugin_pdf(wi, wo, T=1)
    assert torch.equal(p[n - k:], s.plugin_pdf(wi[n - k:].contiguous(), wo[n - k:].contiguous(), T=1))
    del wi, wo, pdf, p
    torch.cuda.empty_cache()
