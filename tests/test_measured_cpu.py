"""CPU tests of the ground-truth evaluator's oracle (oracle/measured_oracle.py).

Mitsuba's `measured` plugin cannot run here (parity unpinned against it); what pins the restatement is
listed in the oracle's header.  Fixture: tests/golden/chm_orange_rgb.bsdf — one of the RGL tensor
files the reference ships under rendering/measuredbsdfs/ (data, 659 KB)."""
import os

import numpy as np
import pytest

from oracle import bsdf_oracle as O
from oracle import measured_oracle as M

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "chm_orange_rgb.bsdf")


@pytest.fixture(scope="module")
def gt():
    return M.MeasuredBSDF(FIXTURE)


def test_tensor_file_reader():
    t = M.read_tensor_file(FIXTURE)
    assert list(t) == ["version", "description", "phi_i", "theta_i", "sigma", "ndf", "vndf", "luminance", "rgb",
                       "jacobian", "valid"]
    assert t["vndf"].shape == (1, 8, 128, 128) and t["rgb"].shape == (1, 8, 3, 32, 32)
    assert t["ndf"].shape == t["sigma"].shape == (2, 128) and t["luminance"].shape == (1, 8, 32, 32)
    assert t["theta_i"].dtype == np.float32 and t["theta_i"][0] == 0 and abs(t["theta_i"][-1] - np.pi / 2) < 1e-6
    assert bytes(t["description"]).decode().startswith("TeckWrap") and tuple(t["version"]) == (1, 0)
    assert t["jacobian"][0] == 1
    with pytest.raises(ValueError, match="not a tensor file"):
        M.read_tensor_file(__file__)


def test_vndf_warp_is_a_consistent_inverse_cdf(gt):
    g = np.random.default_rng(0)
    n = 3000
    u = (g.uniform(size=n), g.uniform(size=n))
    par = (np.zeros(n), g.uniform(0, np.pi / 2, size=n))
    pos, pdf = gt.vndf.sample(u, par)
    (u0, u1), pdf2 = gt.vndf.invert(pos, par)
    assert np.abs(u0 - u[0]).max() < 1e-9 and np.abs(u1 - u[1]).max() < 1e-9
    assert np.allclose(pdf, pdf2, rtol=1e-9) and np.allclose(pdf, gt.vndf.eval(pos, par), rtol=1e-9)
    # the density integrates to one for every incident elevation (midpoint rule on a fine grid)
    k = 512
    xs = (np.arange(k) + 0.5) / k
    X, Y = np.meshgrid(xs, xs)
    for th in (0.0, 0.3, 0.9, 1.5):
        d = gt.vndf.eval((X.ravel(), Y.ravel()), (np.zeros(k * k), np.full(k * k, th)))
        assert abs(d.mean() - 1.0) < 2e-3
    # uniform variates map to samples distributed with that density: mean of 1/pdf over samples = area = 1
    assert abs((1.0 / pdf[pdf > 0]).mean() - 1.0) < 0.15


def test_energy_and_colour(gt):
    g = np.random.default_rng(1)
    n = 100000
    z, ph = g.uniform(size=n), g.uniform(0, 2 * np.pi, size=n)
    r = np.sqrt(1 - z * z)
    wo = np.stack([r * np.cos(ph), r * np.sin(ph), z], 1)
    for th in (0.0, 0.6, 1.2):
        wi = np.tile([np.sin(th), 0.0, np.cos(th)], (n, 1))
        f = gt.eval(wi, wo)
        assert (f >= -0.05).all() and np.isfinite(f).all()
        alb = f.mean(0) * 2 * np.pi          # directional albedo, uniform-hemisphere estimate
        assert (alb > 0.02).all() and (alb < 1.15).all(), alb
        assert alb[0] > 2 * alb[1] > 2 * alb[2] * 0.9  # an orange film: R >> G > B
    # lower hemispheres evaluate to zero (Mitsuba's eval() masks)
    wi = np.tile([0.0, 0.0, 1.0], (4, 1))
    wo4 = np.array([[0, 0, -1.0], [0.6, 0, -0.8], [0, 0, 1.0], [0.6, 0, 0.8]])
    f = gt.eval(wi, wo4)
    assert (f[:2] == 0).all() and (f[2:] > 0).all()
    assert (gt.eval(-wi, wo4) == 0).all()


def test_reciprocity_of_the_isotropic_model(gt):
    """f(wi, wo) = f(wo, wi) holds for the acquisition; the fitted tables reproduce it in the median
    (ratio 1.00) and over orders of magnitude (log-correlation), while individual near-specular pairs of
    this very glossy film differ by a few x (its 2-degree lobe is under-resolved by the 32x32 spectral
    grid; smoother files of the same database agree within 7 %)."""
    g = np.random.default_rng(2)
    n = 4000

    def dirs():
        z, ph = g.uniform(0.3, 1.0, size=n), g.uniform(0, 2 * np.pi, size=n)
        r = np.sqrt(1 - z * z)
        return np.stack([r * np.cos(ph), r * np.sin(ph), z], 1)
    a, b = dirs(), dirs()
    fab = gt.eval(a, b)[:, 0] / b[:, 2]
    fba = gt.eval(b, a)[:, 0] / a[:, 2]
    ok = (fab > 0) & (fba > 0)
    assert ok.mean() > 0.99
    assert abs(np.median(fab[ok] / fba[ok]) - 1.0) < 0.03
    assert np.corrcoef(np.log(fab[ok]), np.log(fba[ok]))[0, 1] > 0.9


def test_agrees_with_the_shipped_network(gt):
    """The reference trains its nets to pdf_disk(x | wi) ∝ lum(f cos)(wo) * clamp(1/cos theta_o, 1, 1e6)
    (learning_repo_cleanup/utils/mitsuba_brdf_scalar.py:85-88): this evaluator and the shipped
    chm_orange_rgb disk net must describe the same lobe."""
    from bsdf_diffusion_sampling_amd import weights as W
    orc = O.Oracle(W.load(W.shipped_path("chm_orange_rgb", "disk")))
    n = 96   # the film's lobe is ~0.05 wide on the disk: a coarser grid under-resolves it
    xs = (np.arange(n) + 0.5) / n * 2 - 1
    X, Y = np.meshgrid(xs, xs)
    m = (X * X + Y * Y) < 0.98
    x = np.stack([X[m], Y[m]], 1)
    wo3 = np.concatenate([x, np.sqrt(1 - (x * x).sum(1, keepdims=True))], 1)
    for wi2 in ([0.4, 0.0], [0.0, 0.7], [-0.5, -0.5]):
        wi2 = np.array(wi2)
        wi3 = np.array([wi2[0], wi2[1], np.sqrt(1 - wi2 @ wi2)])
        f = gt.eval(np.tile(wi3, (len(x), 1)), wo3)
        target = O_lum(f) * np.clip(1 / wo3[:, 2], 1, 1e6)
        p = orc.network_pdf(x, np.tile(wi2, (len(x), 1)), 4)
        assert np.corrcoef(target, p)[0, 1] > 0.9
        assert np.linalg.norm(x[target.argmax()] - x[p.argmax()]) < 0.08
        assert np.linalg.norm(x[target.argmax()] + wi2) < 0.08      # the lobe sits at the mirror direction


def O_lum(rgb):
    return 0.2126 * rgb[:, 0] + 0.7152 * rgb[:, 1] + 0.0722 * rgb[:, 2]


def _field_records(raw):
    """(name -> (ndim, offset of the first u64 shape entry)) of a tensor file's header."""
    import struct
    nf = struct.unpack_from("<I", raw, 14)[0]
    pos, out = 18, {}
    for _ in range(nf):
        nl = struct.unpack_from("<H", raw, pos)[0]
        name = raw[pos + 2: pos + 2 + nl].decode()
        nd = struct.unpack_from("<H", raw, pos + 2 + nl)[0]
        shape_at = pos + 2 + nl + 2 + 1 + 8
        out[name] = (nd, shape_at)
        pos = shape_at + 8 * nd
    return out


def test_native_loader_rejects_malformed_tensor_files(tmp_path):
    """The C loader (bsdfd_measured_create_from_file) validates the shape fields BEFORE any device work: zero or
    oversized dimensions, products that wrap around 2^64, fields that exceed the file, truncated files.  (Runs without a
    GPU: every case fails in the parser.)"""
    import ctypes as C
    import struct
    from bsdf_diffusion_sampling_amd import _lib
    L = _lib.lib()
    raw = open(FIXTURE, "rb").read()
    rec = _field_records(raw)
    assert set(rec) >= {"phi_i", "theta_i", "sigma", "ndf", "vndf", "rgb", "jacobian"}

    def attempt(data, expect):
        p = tmp_path / "bad.bsdf"
        p.write_bytes(bytes(data))
        h = C.c_void_p()
        rc = L.bsdfd_measured_create_from_file(str(p).encode(), C.byref(h))
        msg = L.bsdfd_last_error().decode()
        assert rc == 3 and h.value is None and expect in msg, (rc, msg)   # BSDFD_EIO

    def patched(name, dims):
        b = bytearray(raw)
        nd, at = rec[name]
        assert len(dims) == nd
        struct.pack_into("<%dQ" % nd, b, at, *dims)
        return b

    attempt(patched("theta_i", [0]), "out of range")                          # empty table
    attempt(patched("phi_i", [0]), "out of range")
    attempt(patched("vndf", [1 << 26] * 4), "implausible shape")              # 2^104 elements: the product wraps
    attempt(patched("vndf", [1 << 63, 2, 1, 1]), "implausible shape")         # one absurd dimension
    attempt(patched("rgb", [1, 8, 3, 1 << 24, 1 << 24]), "exceeds the file")   # plausible dimensions, absurd product
    attempt(patched("ndf", [1 << 20, 1 << 20]), "exceeds the file")
    attempt(raw[: len(raw) // 2], "exceeds the file")                          # truncated payload
    attempt(raw[:40], "truncated header")
    attempt(b"not a tensor file at all", "not a tensor file")


def test_against_mitsuba_measured_plugin_when_a_dump_exists(gt):
    """tests/golden/make_mitsuba_golden.py writes `mitsuba_measured_eval.npz` on a machine where Mitsuba 3 is installed
    (Mitsuba's `measured` plugin = what the reference's eval() delegates to, rendering/brdf_measured_disk.py:36-42,103-110).
    No such machine has existed for this build, so this test SKIPS and SURVEY §8 row f3 stays parity-unpinned; the day the
    file is committed the oracle is held to it."""
    path = os.path.join(HERE, "golden", "mitsuba_measured_eval.npz")
    if not os.path.exists(path):
        pytest.skip("no Mitsuba dump (tests/golden/make_mitsuba_golden.py needs a machine with `pip install mitsuba`)")
    d = np.load(path)
    f = gt.eval(d["wi"].astype(np.float64), d["wo"].astype(np.float64))
    ref = d["f_cos"].astype(np.float64)
    scale = np.percentile(ref.max(1), 99)
    assert np.percentile(np.abs(f - ref).max(1) / scale, 99) < 1e-3
