"""Pin the oracle (oracle/bsdf_oracle.py) to OUTPUTS OF THE REFERENCE ITSELF.

The goldens under tests/golden/*.npz were produced by tests/golden/make_golden.py,
which imports the reference's rendering/utils/{model,mlp_brdf_sampling}.py
unmodified.  Bounds: the fp64 oracle must agree with the reference's own fp64 run
to round-off (1e-10), and with the reference's fp32 run to the fp32 noise floor
measured in BASELINE.md §2 (median ~2e-6, p99 <= 2e-4 incl. near-singular rows).
"""
import os

import numpy as np
import pytest

from oracle import bsdf_oracle as O


def rel(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-30)


def test_positional_encoding_rows(golden_case):
    _, g, _ = golden_case
    wi = g["wi"][:64].astype(np.float64)
    assert np.abs(O.positional_encoding(wi, 5) - g["pe5_rows"]).max() < 5e-7
    assert np.abs(O.positional_encoding(wi, 3) - g["pe3_rows"]).max() < 5e-7


def test_sampling_matches_reference_fp64(golden_case):
    _, g, fw = golden_case
    orc = O.Oracle(fw)
    x, p = orc.network_sampling(g["wi"], g["x0"], int(g["meta_T"]))
    assert np.abs(x - g["sample_x_f64"]).max() < 1e-10
    assert rel(p, g["sample_pdf_f64"]).max() < 1e-9
    assert np.array_equal(np.sign(p), np.sign(g["sample_pdf_f64"]))


@pytest.mark.parametrize("stem", ["chm_orange_rgb_spherical_n16k", "aniso_miro_7_rgb_spherical_complex_n16k"])
def test_large_fixtures_pin_the_oracle_too(stem):
    """The 16 384-row fixtures of the two hard cases (make_golden.py --large): the oracle reproduces the reference's fp64 run
    to round-off and its fp32 runs (sample, pdf at produced and at fresh points) to fp32 noise — same bounds as the small ones."""
    from conftest import load_case
    g, fw = load_case(stem)
    assert g["wi"].shape[0] == 16384
    orc = O.Oracle(fw)
    T = int(g["meta_T"])
    x, p, acc = orc.network_sampling(g["wi"], g["x0"], T, return_acc=True)
    assert np.abs(x - g["sample_x_f64"]).max() < 1e-10
    assert rel(p, g["sample_pdf_f64"]).max() < 1e-9
    assert np.abs(x - g[f"sample_x_T{T}"]).max() < 1e-4
    r = rel(p, g[f"sample_pdf_T{T}"])[np.abs(1.0 / acc) > 1e-3]
    assert np.median(r) < 1e-5 and np.percentile(r, 99) < 2e-4
    for which in "ab":
        pp = orc.network_pdf(g[f"pdf_wo_{which}"], g["wi"], T)
        ref = g[f"pdf_{which}_T{T}"]
        big = np.abs(ref) > 1e-6 * np.abs(ref).max()
        r = rel(pp, ref)[big]
        assert np.median(r) < 1e-5 and np.percentile(r, 99) < 1e-3
        assert np.mean(np.sign(pp[big]) == np.sign(ref[big])) > 0.999


@pytest.mark.parametrize("T", [1, 4, 8])
def test_sampling_matches_reference_fp32(golden_case, T):
    _, g, fw = golden_case
    if f"sample_x_T{T}" not in g.files:
        pytest.skip("T not in fixture")
    orc = O.Oracle(fw)
    x, p = orc.network_sampling(g["wi"], g["x0"], T)
    assert np.abs(x - g[f"sample_x_T{T}"]).max() < 5e-5
    # error metric of SURVEY.md §8(d): rows with |prod det J| > 1e-3 (the fp32
    # reference itself loses digits where a step's det J ~ 0; T=1 is the worst)
    _, acc = orc.flow(g["x0"], g["wi"], T, reverse=False)
    r = rel(p, g[f"sample_pdf_T{T}"])[np.abs(1.0 / acc) > 1e-3]
    assert np.median(r) < 1e-5 and np.percentile(r, 99) < (2e-3 if T == 1 else 2e-4)


@pytest.mark.parametrize("which", ["a", "b"])
def test_pdf_matches_reference_fp32(golden_case, which):
    _, g, fw = golden_case
    orc = O.Oracle(fw)
    for T in (4, 8):
        if f"pdf_{which}_T{T}" not in g.files:
            continue
        p = orc.network_pdf(g[f"pdf_wo_{which}"], g["wi"], T)
        ref = g[f"pdf_{which}_T{T}"]
        big = np.abs(ref) > 1e-6 * np.abs(ref).max()  # rows whose density is resolved in fp32
        r = rel(p, ref)[big]
        assert np.median(r) < 1e-5 and np.percentile(r, 99) < 1e-3
        # sign of det J is kept: same sign wherever |pdf| is resolved
        assert np.mean(np.sign(p[big]) == np.sign(ref[big])) > 0.999


def test_single_step_velocity_and_jacobian(golden_case):
    _, g, fw = golden_case
    orc = O.Oracle(fw)
    pe = O.positional_encoding(g["wi"][:64].astype(np.float64), 5)
    for k in (0, 1):
        v, d0, d1 = orc.velocity_jacobian(g["x0"][:64].astype(np.float64), float(g[f"step{k}_alpha"]), pe)
        scale = 1.0 + np.abs(g[f"step{k}_g0"]).max() + np.abs(g[f"step{k}_g1"]).max()
        assert np.abs(v - g[f"step{k}_v"]).max() < 2e-5
        # reference rows are gradients of v_0 and v_1 (mlp_brdf_sampling.py:31-41)
        assert np.abs(np.stack([d0[:, 0], d1[:, 0]], 1) - g[f"step{k}_g0"]).max() < 2e-5 * scale
        assert np.abs(np.stack([d0[:, 1], d1[:, 1]], 1) - g[f"step{k}_g1"]).max() < 2e-5 * scale


def test_base_density(golden_case):
    _, g, fw = golden_case
    orc = O.Oracle(fw)
    assert np.abs(orc.base_forward(g["wi"]) - g["base_fwd"]).max() < 2e-5
    lp = orc.base_log_prob(g["x0"], g["wi"])
    assert np.abs(lp - g["base_logp_x0"]).max() < 2e-4 * (1 + np.abs(g["base_logp_x0"]).max() / 10)


def test_von_mises_log_prob_kappa_sweep():
    import os
    from conftest import GOLDEN
    k = np.load(os.path.join(GOLDEN, "von_mises_kappa_sweep.npz"))
    kap = k["kappa"].astype(np.float64)[:, None]
    lp = kap * np.cos(k["phi"][None, :] - float(k["mu"])) - np.log(2 * np.pi) - O.log_i0(kap)
    assert np.abs(lp - k["logp"]).max() < 2e-4  # fp32 torch at kappa=1e3 carries ~1e-4 abs


def test_oracle_fp32_mode_close_to_fp64(golden_case):
    _, g, fw = golden_case
    T = int(g["meta_T"])
    x64, p64 = O.Oracle(fw).network_sampling(g["wi"], g["x0"], T)
    x32, p32 = O.Oracle(fw, np.float32).network_sampling(g["wi"], g["x0"], T)
    assert x32.dtype == np.float32
    assert np.abs(x32 - x64).max() < 1e-4
    assert np.median(rel(p32, p64)) < 1e-5


def test_toy_1d_config1():
    """BASELINE.json configs[0]: 1-D toy flow, CPU plumbing only."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "toy_1d.npz"))
    params = [(g[f"W{i}"].astype(np.float64), g[f"b{i}"].astype(np.float64)) for i in range(5)]
    x, acc = O.toy_flow_1d(params, g["x0"], T=8)
    assert np.abs(x - g["xT"]).max() < 2e-5
    assert rel(acc, g["acc"]).max() < 2e-4


def test_plugin_guards_disk_and_spherical():
    """Plugin-level post-processing restated from brdf_measured_{disk,spherical}.py."""
    from conftest import load_case
    g, fw = load_case("aniso_miro_7_rgb_disk")
    orc = O.Oracle(fw)
    wi2 = g["wi"].astype(np.float64)
    wi3 = np.concatenate([wi2, np.sqrt(np.maximum(1 - (wi2 ** 2).sum(1), 0))[:, None]], 1)
    x0 = g["x0"].astype(np.float64).copy()
    x0[:8] *= 50.0  # drive some samples outside the disk
    wo3, pdf = O.plugin_sample_disk(orc, wi3, x0, T=4)
    r2 = wo3[:, 0] ** 2 + wo3[:, 1] ** 2
    assert np.all(r2 < 0.995)
    bad = (wo3[:, 0] == 0) & (wo3[:, 1] == 0)
    assert bad.sum() >= 1 and np.all(pdf[bad] == 0) and np.all(wo3[bad, 2] == 1)
    assert np.allclose((wo3 ** 2).sum(1), 1.0)
    p = O.plugin_pdf_disk(orc, wi3, wo3, T=4)
    assert np.all(np.isfinite(p))
    g, fw = load_case("aniso_miro_7_rgb_spherical")
    orc = O.Oracle(fw)
    th, ph = g["wi"][:, 0].astype(np.float64), g["wi"][:, 1].astype(np.float64)
    wi3 = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)], 1)
    assert np.abs(O.cart_to_spher(wi3) - g["wi"]).max() < 1e-4  # acos(z/(r+1e-8)) near theta=0
    wo3, pdf = O.plugin_sample_spherical(orc, wi3, g["x0"], T=8)
    assert np.allclose((wo3 ** 2).sum(1), 1.0)
    assert np.all(pdf[wo3[:, 2] <= 0] == 0)


@pytest.mark.parametrize("T", [128, 256])
def test_disk_reflow_teacher_matches_reference(T):
    """SURVEY §8 f2, disk teacher (learning_repo_cleanup/disk_domain_sampling.py:93-110: 32 x 3 diffusion net, T = 128 /
    the script's default 256 Euler steps, no Jacobian): the oracle's samples-only flow vs the reference's own
    `NN_cond_pos_simpler` stepped by tests/golden/make_teacher_golden.py — fp64 to round-off, fp32 to its noise."""
    import os
    from conftest import GOLDEN
    from bsdf_diffusion_sampling_amd import weights as W
    g = np.load(os.path.join(GOLDEN, "disk_teacher_aniso_miro_7_rgb.npz"))
    fw = W.load(W.shipped_path("aniso_miro_7_rgb", "disk", "diffusion"))
    assert (fw.width, fw.n_hidden) == (32, 3)
    x, _ = O.Oracle(fw).flow(g["x0"], g["wi"], T, reverse=False)
    assert np.abs(x - g[f"x_T{T}_f64"]).max() < 1e-10
    assert np.abs(x - g[f"x_T{T}_f32"]).max() < 5e-5


PLUGIN_CASES = ["aniso_miro_7_rgb_disk", "chm_orange_rgb_disk", "vch_silk_blue_rgb_disk", "aniso_miro_7_rgb_spherical",
                "chm_orange_rgb_spherical", "bsdf_3_spherical"]


@pytest.mark.parametrize("stem", PLUGIN_CASES + ["chm_orange_rgb_spherical_n16k"])
def test_plugin_level_oracle_vs_reference_plugin_goldens(stem):
    """SURVEY.md §8(c) last row: plugin-level (wo3, pdf_sa) after guards.  tests/golden/<stem>_plugin.npz holds the
    reference's own operators followed by the plugins' tensor ops (make_plugin_golden.py names the lines); the oracle's
    plugin-level restatement must reproduce the reference's fp64 run to round-off and its fp32 run to fp32 noise."""
    from conftest import GOLDEN, load_case
    _, fw = load_case(stem)
    p = np.load(os.path.join(GOLDEN, stem + "_plugin.npz"))
    orc = O.Oracle(fw)
    T, full = int(p["meta_T"]), bool(p["meta_full_sphere"])
    if fw.domain == 0:
        wo_o, pdf_o = O.plugin_sample_disk(orc, p["wi3"], p["x0"], T=T)
        pl_o = O.plugin_pdf_disk(orc, p["pdf_wi3"], p["pdf_wo3"], T=T)
        ps_o = O.plugin_pdf_disk(orc, p["wi3"], p["sample_wo3"], T=T)
    else:
        wo_o, pdf_o = O.plugin_sample_spherical(orc, p["wi3"].astype(np.float64), p["x0"], T=T, full_sphere=full)
        pl_o = O.plugin_pdf_spherical(orc, p["pdf_wi3"].astype(np.float64), p["pdf_wo3"].astype(np.float64), T=T, full_sphere=full)
        ps_o = O.plugin_pdf_spherical(orc, p["wi3"].astype(np.float64), p["sample_wo3"].astype(np.float64), T=T, full_sphere=full)
    # fp64 run of the reference + the plugin ops in fp64: round-off agreement, zeros (guards) in the same rows
    assert np.abs(wo_o - p["sample_wo3_f64"]).max() < 1e-12
    ref = p["sample_pdf_sa_f64"]
    assert np.array_equal(ref == 0, pdf_o == 0)
    nz = ref != 0
    assert (np.abs(pdf_o - ref)[nz] / np.abs(ref[nz])).max() < 1e-9
    if fw.domain == 0:  # the guard was exercised: r^2 >= 0.995 rows give wo = (0, 0, 1), pdf = 0
        bad = (p["sample_wo3"][:, 0] == 0) & (p["sample_wo3"][:, 1] == 0)
        assert bad.sum() >= 1 and np.all(p["sample_pdf_sa"][bad] == 0) and np.all(pdf_o[bad] == 0)
    # fp32 run (what the plugins really compute): fp32 noise only
    scale = np.percentile(np.abs(ref), 99)
    ok = np.abs(ref) > 1e-6 * scale
    rel32 = np.abs(p["sample_pdf_sa"] - ref)[ok] / np.abs(ref[ok])
    assert np.median(rel32) < 1e-5 and np.percentile(rel32, 99) < 3e-4
    assert np.percentile(np.abs(p["sample_wo3"] - p["sample_wo3_f64"]), 99) < 1e-5
    # pdf(): the reference forces fp32 (mlp_brdf_sampling.py:71,146) — compare at fp32 noise; masked lanes are exactly 0
    for got, want in ((p["pdf_sa"], pl_o), (p["pdf_sa_of_samples"], ps_o)):
        okp = np.abs(want) > 1e-6 * np.percentile(np.abs(want), 99)
        r = np.abs(got - want)[okp] / np.abs(want[okp])
        assert np.median(r) < 2e-5 and np.percentile(r, 99) < 3e-4
    if not full:
        assert np.all(p["pdf_sa"][:16] == 0) and np.all(pl_o[:16] == 0)   # cos(theta_o) <= 0 / cos(theta_i) <= 0 lanes
