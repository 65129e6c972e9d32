"""BASELINE.json configs[3] and configs[4] AT THEIR STATED SIZES on one GPU (the 8-GPU forms need the driver's node;
what an 8-GPU run adds on top of these is the RCCL hop of the final concatenation only).

config 4: "All paper measured BSDFs, 128M mixed queries sharded across 8 x MI355X" — here the 128 Mi wavefront is served
          once whole and once as 8 contiguous virtual shards of 16 Mi (each shard bucketed on its own, Philox offset =
          the shard's first lane): the concatenation must equal the whole call bit for bit, i.e. a rank's results do not
          depend on how many ranks there are.  Oracle on a 4 096-row subsample across >= 8 materials.
config 5: "Full Mitsuba matpreview scene at 1024 spp ... image-tile split" — the reference's driver loop
          (rendering/brdf_measured_disk.py:146-155: 256 passes of mi.render(spp=4); film 512 x 512,
          rendering/matpreview/scene_measured.xml:2-4) through the wavefront harness, whole film vs 8 row tiles: the
          films must be bit-equal.
"""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from conftest import load_case  # noqa: E402
from oracle import bsdf_oracle as O  # noqa: E402


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    return torch.device("cuda", 0)


def test_config4_128Mi_mixed_queries_in_8_virtual_shards():
    from bsdf_diffusion_sampling_amd.materials import MaterialTable
    dev = _dev()
    t0 = time.perf_counter()
    tab = MaterialTable.all_measured()
    assert len(tab) == 52                                        # 27 disk + 25 spherical measured materials
    n, shards = 1 << 27, 8
    g = torch.Generator(device=dev).manual_seed(4)
    ids = torch.randint(0, len(tab), (n,), generator=g, device=dev)
    u = torch.rand((n, 2), generator=g, device=dev)
    th, ph = 1.5 * u[:, 0], (2 * u[:, 1] - 1) * np.pi           # upper-hemisphere unit vectors serve both domains
    wi = torch.stack([torch.sin(th) * torch.cos(ph), torch.sin(th) * torch.sin(ph), torch.cos(th)], 1).contiguous()
    del u, th, ph
    # the whole wavefront in one call (bucketing + 2 segmented launches), in-kernel Philox RNG keyed by the lane index
    wo, pdf = tab.sample(ids, wi, seed=77, offset=0)
    p = tab.pdf(ids, wi, wo)
    assert torch.isfinite(wo).all() and torch.isfinite(pdf).all() and torch.isfinite(p).all()
    assert torch.allclose((wo * wo).sum(1), torch.ones(n, device=dev), atol=1e-4)
    # 8 contiguous shards, each bucketed on its own = what 8 ranks do (bench.py --workload mixed_16Mi --gpus 8)
    per = n // shards
    for r in range(shards):
        a, b = r * per, (r + 1) * per
        wo_r, pdf_r = tab.sample(ids[a:b], wi[a:b], seed=77, offset=a)
        assert torch.equal(wo_r, wo[a:b]) and torch.equal(pdf_r, pdf[a:b]), f"shard {r} differs from the whole call"
        assert torch.equal(tab.pdf(ids[a:b], wi[a:b], wo_r), p[a:b])
        del wo_r, pdf_r
    # oracle on a subsample: 4 096 rows spread over 8 materials (4 disk, 4 spherical), injected base draws
    mats = [0, 5, 13, 26, 27, 33, 41, 51]
    rng = np.random.default_rng(11)
    checked = 0
    for m in mats:
        rows = (ids[: 1 << 22] == m).nonzero()[:512, 0]
        assert rows.numel() == 512
        fw = tab.samplers[m].weights
        orc, orc32 = O.Oracle(fw), O.Oracle(fw, np.float32)
        wis = wi[rows].contiguous()
        wi64 = wis.cpu().numpy().astype(np.float64)
        T = tab.T[m]
        if fw.domain == 0:
            cond = wi64[:, :2]
            x0 = orc.base_sample(cond, rng.standard_normal((512, 2))).astype(np.float32)
            wo_o, pdf_o = O.plugin_sample_disk(orc, wi64, x0.astype(np.float64), T=T)
            wo_32, pdf_32 = O.plugin_sample_disk(orc32, wis.cpu().numpy(), x0, T=T)
        else:
            cond = O.cart_to_spher(wi64)
            mu, kappa = orc.base_von_mises_params(cond)
            x0 = orc.base_sample(cond, rng.standard_normal(512), phi=rng.vonmises(mu, kappa)).astype(np.float32)
            wo_o, pdf_o = O.plugin_sample_spherical(orc, wi64, x0.astype(np.float64), T=T)
            wo_32, pdf_32 = O.plugin_sample_spherical(orc32, wis.cpu().numpy(), x0, T=T)
        idm = torch.full((512,), m, dtype=torch.int64, device=dev)
        wo_s, pdf_s = tab.sample(idm, wis, x0=torch.from_numpy(x0).to(dev))        # through the table's segmented path
        _, acc = orc.flow(x0.astype(np.float64), cond, T, reverse=False)
        sel = (np.abs(acc) > 1e-3) & (np.abs(acc) < 1e3) & (np.abs(pdf_o) > 1e-6 * np.percentile(np.abs(pdf_o), 99))
        err_wo, noise_wo = np.abs(wo_s.cpu().numpy() - wo_o), np.abs(wo_32.astype(np.float64) - wo_o)
        rel = np.abs(pdf_s.cpu().numpy() - pdf_o)[sel] / np.abs(pdf_o[sel])
        noise = np.abs(pdf_32.astype(np.float64) - pdf_o)[sel] / np.abs(pdf_o[sel])
        assert np.percentile(err_wo, 99) <= 1e-5 and err_wo.max() <= 1e-4, tab.stems[m]
        assert np.percentile(rel, 99) <= 1e-4, (tab.stems[m], np.percentile(rel, 99), np.percentile(noise, 99))
        checked += 512
    assert checked == 4096
    torch.cuda.synchronize()
    print(f"config 4 at full size: {time.perf_counter() - t0:.1f} s")


def test_config5_matpreview_1024spp_as_8_row_tiles():
    from bsdf_diffusion_sampling_amd import wavefront as WF
    from bsdf_diffusion_sampling_amd.brdf_measured_disk import MyBSDF
    _dev()
    t0 = time.perf_counter()
    plug = MyBSDF({"filename": "aniso_miro_7_rgb", "measured": False})
    r = WF.WavefrontRenderer(plug, WF.Camera(width=512, height=512))
    passes, spp, tiles = 256, 4, 8                               # brdf_measured_disk.py:146-155: 256 x mi.render(spp=4) = 1024 spp
    whole = r.render(passes=passes, spp=spp, seed=3)
    rows = 512 // tiles
    parts = [r.render(passes=passes, spp=spp, seed=3, rows=(k * rows, (k + 1) * rows)) for k in range(tiles)]
    assert torch.equal(torch.cat(parts, 0), whole)               # a tile-split render is the same image, bit for bit
    img = whole.cpu().numpy()
    assert np.isfinite(img).all() and img.min() >= 0 and 0.01 < img.mean() < 10
    ball = img[200:312, 200:312].mean()
    assert ball > 0                                              # the material ball is lit
    torch.cuda.synchronize()
    print(f"config 5 at full size (2.7e8 paths, whole + 8 tiles): {time.perf_counter() - t0:.1f} s")


def test_wavefront_pipeline_equals_the_serial_stages():
    """materials.WavefrontPipeline (bucket + gather of wavefront k+1 and scatter of wavefront k-1 on side streams under the flow
    kernels of wavefront k — what bench.py's mixed_16Mi workload runs): five independent wavefronts with different material
    ids give, bit for bit, what bucket -> gather -> sample -> pdf -> scatter give one after the other on one stream; buffers
    are recycled after two wavefronts; lanes of the extra bins come back as zeros."""
    import torch
    from bsdf_diffusion_sampling_amd.materials import MaterialTable, WavefrontPipeline
    dev = _dev()
    tab = MaterialTable(["aniso_miro_7_rgb_disk", "chm_orange_rgb_disk", "aniso_miro_7_rgb_spherical", "chm_orange_rgb_spherical",
                         "bsdf_3_spherical"])
    n = 300_001
    g = torch.Generator().manual_seed(9)
    z = 0.05 + 0.9 * torch.rand(n, generator=g)
    ph = 6.2831853 * torch.rand(n, generator=g)
    wi = torch.stack([torch.sqrt(1 - z * z) * torch.cos(ph), torch.sqrt(1 - z * z) * torch.sin(ph), z], 1).float().to(dev)
    pipe = WavefrontPipeline(tab)
    ctx_p, ctx_s = {}, {}
    prev = None
    for k in range(5):
        extra = 1 if k == 3 else 0
        ids = torch.randint(0, len(tab) + extra, (n,), generator=torch.Generator().manual_seed(100 + k)).to(dev)
        wi_k = wi.roll(k, 0).contiguous()
        # the pipeline reads its inputs on a side stream and orders that stream behind their producer itself: by
        # default behind everything enqueued on the calling stream so far — NO host synchronisation here, the roll / copy above
        # is still in flight; with an explicit event of the producer; ready=False only for inputs that are complete (k == 4)
        if k == 2:
            ev = torch.cuda.current_stream().record_event()
            w = pipe.push(ids, wi_k, seed=40 + k, offset=7 * k, ctx=ctx_p, extra_bins=extra, ready=ev)
        elif k == 4:
            torch.cuda.synchronize()
            w = pipe.push(ids, wi_k, seed=40 + k, offset=7 * k, ctx=ctx_p, extra_bins=extra, ready=False)
        else:
            w = pipe.push(ids, wi_k, seed=40 + k, offset=7 * k, ctx=ctx_p, extra_bins=extra)
        # the same wavefront, stage by stage
        plan = tab.bucket(ids, extra)
        wi_b = tab.gather(plan, wi_k)
        wo_b, pdf_b = tab.sample(plan, wi_b, seed=40 + k, offset=7 * k, bucketed=True, ctx=ctx_s)
        p_b = tab.pdf(plan, wi_b, wo_b, bucketed=True, ctx=ctx_s)
        ref = tab.scatter(plan, wo_b, pdf_b, p_b)
        got = w.result()
        torch.cuda.synchronize()
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
        if extra:
            assert (got[1][ids == len(tab)] == 0).all() and (got[0][ids == len(tab)] == 0).all()
        assert torch.isfinite(got[2]).all() and (got[1] > 0).float().mean() > 0.5
        if prev is not None:
            assert prev is not w                       # two slots, alternating
        prev = w
