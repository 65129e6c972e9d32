"""CPU tests of the wavefront-harness oracle (oracle/wavefront_oracle.py) and host logic."""
import numpy as np
import pytest

from oracle import wavefront_oracle as WO


def _scene(w=64, h=48):
    from bsdf_diffusion_sampling_amd.wavefront import Camera
    import math
    cam = Camera(width=w, height=h)
    r, u, f = cam.basis()
    return dict(origin=cam.origin, right=r, up=u, forward=f, tan_half_fov=math.tan(math.radians(cam.fov_deg) / 2),
                width=w, height=h, center=(0, 0, 0), radius=1.0, albedo=(1.0, 1.0, 1.0))


def test_philox_known_answer_vectors():
    """Random123 kat_vectors, philox4x32 with 10 rounds."""
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        got = WO.philox4x32(key[0], key[1], *ctr)
        assert tuple(int(g) for g in got) == want


def test_mis_power_heuristic():
    a = np.array([0.0, 1.0, 2.0, 1e12, 3.0], dtype=np.float32)
    b = np.array([1.0, 1.0, 1.0, 1.0, 0.0], dtype=np.float32)
    w = WO.mis_power(a, b)
    assert np.allclose(w, [0.0, 0.5, 0.8, 1.0, 1.0])
    assert np.allclose(WO.mis_power(a[1:3], b[1:3]) + WO.mis_power(b[1:3], a[1:3]), 1.0)


def test_env_lookup_texel_centres_and_wrap():
    g = np.random.default_rng(0)
    env = g.uniform(size=(8, 16, 3)).astype(np.float32)
    h, w = env.shape[:2]
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    th, ph = np.pi * (yy.ravel() + 0.5) / h, 2 * np.pi * (xx.ravel() + 0.5) / w
    d = np.stack([np.sin(th) * np.sin(ph), np.cos(th), -np.sin(th) * np.cos(ph)], 1).astype(np.float32)
    got = WO.env_lookup(env, d)
    assert np.abs(got - env.reshape(-1, 3)).max() < 2e-4
    const = np.full((4, 8, 3), 0.7, dtype=np.float32)
    dd = g.normal(size=(1000, 3)); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    assert np.allclose(WO.env_lookup(const, dd.astype(np.float32)), 0.7, atol=1e-6)


def test_primary_rays_geometry():
    sc = _scene()
    wi, wl, nrm, d = WO.primary(sc, 0, sc["height"], 2, seed=3, pass_idx=1)
    n = sc["width"] * sc["height"] * 2
    assert wi.shape == wl.shape == nrm.shape == d.shape == (n, 3)
    assert np.allclose((d * d).sum(1), 1, atol=1e-5) and np.allclose((wl * wl).sum(1), 1, atol=1e-5)
    hit = (nrm != 0).any(1)
    assert 0.1 < hit.mean() < 0.9
    assert np.allclose((nrm[hit] ** 2).sum(1), 1, atol=1e-5) and np.allclose((wi ** 2).sum(1), 1, atol=1e-4)
    assert (wi[hit, 2] > 0).all()                      # the camera sees front faces
    assert np.allclose(wi[hit, 2], -(d[hit] * nrm[hit]).sum(1), atol=1e-5)
    assert (wi[~hit] == [0, 0, 1]).all()
    assert abs(wl[:, 2].mean() - 2 / 3) < 0.01          # cosine-weighted: E[cos] = 2/3
    # hit points lie on the sphere: |o + t d| = 1 with n = hit point
    # different passes / seeds decorrelate; same arguments reproduce
    again = WO.primary(sc, 0, sc["height"], 2, seed=3, pass_idx=1)
    assert all(np.array_equal(a, b) for a, b in zip((wi, wl, nrm, d), again))
    other = WO.primary(sc, 0, sc["height"], 2, seed=3, pass_idx=2)
    assert not np.array_equal(other[3], d)
    # a row range is a slice of the full frame (RNG keyed by the global path index)
    part = WO.primary(sc, 10, 20, 2, seed=3, pass_idx=1)
    lo, hi = 10 * sc["width"] * 2, 20 * sc["width"] * 2
    assert all(np.array_equal(a[lo:hi], b) for a, b in zip((wi, wl, nrm, d), part))


def test_shade_white_furnace_and_mis_partition():
    """Constant environment E, albedo a: misses see E; a hit with pdf_bsdf == pdf_light (a Lambertian
    'net') gets weights 1/2 + 1/2 -> exactly a * E per path."""
    sc = _scene(16, 8)
    sc["albedo"] = (0.5, 0.25, 1.0)
    env = np.full((4, 8, 3), 2.0, dtype=np.float32)
    wi, wl, nrm, d = WO.primary(sc, 0, 8, 4, seed=0, pass_idx=0)
    g = np.random.default_rng(1)
    u = g.uniform(size=(len(wi), 2))
    r, ph = np.sqrt(u[:, 0]), 2 * np.pi * u[:, 1]
    wo = np.stack([r * np.cos(ph), r * np.sin(ph), np.sqrt(1 - u[:, 0])], 1).astype(np.float32)
    pdf_o, pdf_l = (wo[:, 2] / np.pi).astype(np.float32), (wl[:, 2] / np.pi).astype(np.float32)
    img = WO.shade(sc, env, 4, wo, pdf_o, wl, pdf_l, nrm, d)
    hit = (nrm != 0).any(1).reshape(-1, 4)
    full = hit.all(1)
    assert np.allclose(img[full], 2.0 * np.array(sc["albedo"]), rtol=1e-5)
    assert np.allclose(img[~hit.any(1)], 2.0, rtol=1e-5)
    # invalid BSDF samples (pdf 0 / nan) contribute nothing from the BSDF strategy
    pdf_bad = pdf_o.copy(); pdf_bad[::2] = 0; pdf_bad[1::4] = np.nan
    img2 = WO.shade(sc, env, 4, wo, pdf_bad, wl, pdf_l, nrm, d)
    assert np.isfinite(img2).all() and (img2[full] <= img[full] + 1e-6).all()


def test_camera_basis_and_row_sharding():
    from bsdf_diffusion_sampling_amd.wavefront import Camera, make_sky
    from bsdf_diffusion_sampling_amd.sharding import shard_range
    r, u, f = Camera(origin=(1, 2, 3), target=(0, 0.5, 0)).basis()
    m = np.stack([r, u, f])
    assert np.allclose(m @ m.T, np.eye(3), atol=1e-12) and np.allclose(np.cross(r, u), -f, atol=1e-12)
    rows = [shard_range(512, k, 8) for k in range(8)]
    assert rows[0][0] == 0 and rows[-1][1] == 512 and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
    sky = make_sky(32, 64, seed=1)
    assert sky.shape == (32, 64, 3) and bool((sky >= 0).all()) and float(sky.max()) > 1.0


def test_png_writer_roundtrip(tmp_path):
    """render_cli.write_png is the stdlib-only stand-in for mi.util.write_bitmap (brdf_measured_disk.py:158)."""
    import struct
    import zlib
    from bsdf_diffusion_sampling_amd.render_cli import tonemap, write_png
    g = np.random.default_rng(0)
    img = g.uniform(0, 4, size=(5, 7, 3))
    rgb8 = tonemap(img)
    assert rgb8.dtype == np.uint8 and rgb8.shape == (5, 7, 3) and rgb8.max() <= 255
    assert (tonemap(np.zeros((1, 1, 3))) == 0).all() and (np.diff(tonemap(np.linspace(0, 10, 50)[None, :, None].repeat(3, 2))[0, :, 0].astype(int)) >= 0).all()
    p = tmp_path / "x.png"
    write_png(str(p), rgb8)
    raw = p.read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, {}
    while pos < len(raw):
        n, tag = struct.unpack(">I", raw[pos:pos + 4])[0], raw[pos + 4:pos + 8]
        data = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + data) & 0xFFFFFFFF
        chunks[tag] = data
        pos += 12 + n
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[b"IHDR"][:10])
    assert (w, h, depth, ctype) == (7, 5, 8, 2)
    rows = zlib.decompress(chunks[b"IDAT"])
    back = np.frombuffer(rows, dtype=np.uint8).reshape(5, 1 + 7 * 3)[:, 1:].reshape(5, 7, 3)
    assert np.array_equal(back, rgb8)


def test_array_scene_oracle_geometry():
    """Ball arrays over a checkerboard floor (the shape of matpreview/disney_bsdf_array*.xml)."""
    from bsdf_diffusion_sampling_amd.wavefront import ARRAY0_LAYOUT, ARRAY0_MATERIALS, array0_scene
    import math
    cam, centers, radii = array0_scene(96, 72)
    assert len(centers) == len(ARRAY0_LAYOUT) == len(ARRAY0_MATERIALS) == 12
    r, u, f = cam.basis()
    sc = dict(origin=cam.origin, right=r, up=u, forward=f, tan_half_fov=math.tan(math.radians(cam.fov_deg) / 2),
              width=cam.width, height=cam.height, spheres=list(zip(centers, radii)),
              plane=dict(y=0.0, c0=0.4, c1=0.2, scale=2.0), albedo=(1.0, 1.0, 1.0))
    wi, wl, nrm, d, mat = WO.primary(sc, 0, cam.height, 1, seed=1, pass_idx=0, with_material=True)
    n_b = len(centers)
    assert set(np.unique(mat)) <= set(range(n_b + 2))
    assert len(np.unique(mat[mat < n_b])) == 12          # the camera sees all balls
    floor, miss, ball = mat == n_b, mat == n_b + 1, mat < n_b
    assert floor.mean() > 0.2 and ball.mean() > 0.1 and floor.sum() + miss.sum() + ball.sum() == len(mat)
    sc_up = dict(sc, forward=np.array([0.0, 1.0, 0.0]), right=np.array([1.0, 0.0, 0.0]), up=np.array([0.0, 0.0, 1.0]))
    assert (WO.primary(sc_up, 0, 4, 1, 1, 0, with_material=True)[4] == n_b + 1).all()      # looking up: only sky
    assert (nrm[floor] == [0, 1, 0]).all() and (nrm[miss] == 0).all()
    assert set(np.unique(wi[floor])) == {np.float32(0.4), np.float32(0.2)}      # the two checker reflectances
    assert np.allclose((nrm[ball] ** 2).sum(1), 1, atol=1e-5) and (wi[ball, 2] > 0).all()
    # the hit point of a ball path lies on that ball
    o = np.asarray(cam.origin)
    for k in range(n_b):
        sel = mat == k
        if sel.any():
            p = np.asarray(centers[k]) + radii[k] * nrm[sel].astype(np.float64)
            dirs = (p - o) / np.linalg.norm(p - o, axis=1, keepdims=True)
            assert np.abs(dirs - d[sel]).max() < 1e-4
    # shading: constant environment E -> floor pixels are reflectance * E, misses E
    env = np.full((4, 8, 3), 1.5, dtype=np.float32)
    z = np.zeros_like(wl)
    img = WO.shade(sc, env, 1, z, np.zeros(len(wl), np.float32), wl, np.zeros(len(wl), np.float32), nrm, d,
                   wi=wi, material=mat)
    assert np.allclose(img[floor], 1.5 * wi[floor, :1], rtol=1e-5) and np.allclose(img[miss], 1.5, rtol=1e-5)


def test_parse_reference_scene_file(tmp_path):
    """The scene-file reader takes the `mybsdf` shapes of a reference-format Mitsuba XML (layout of
    rendering/matpreview/disney_bsdf_array*.xml) and nothing else."""
    from bsdf_diffusion_sampling_amd.wavefront import parse_matpreview_xml, scene_from_matpreview_xml
    xml = """<?xml version="1.0" encoding="utf-8"?>
<scene version="0.5.0">
  <bsdf type="diffuse" id="__diffmat"><rgb name="reflectance" value="0.18 0.18 0.18"/></bsdf>
  <shape type="serialized" id="Shell0">
    <transform name="toWorld"><scale x="0.5" y="0.5" z="0.5"/><translate z="0.01"/><translate x="-4" y="1"/></transform>
    <bsdf type="mybsdf"><string name='filename' value='chm_orange_rgb'/></bsdf>
  </shape>
  <shape type="serialized" id="Interior0">
    <transform name="toWorld"><translate x="-4" y="1"/></transform><ref name="bsdf" id="__diffmat"/>
  </shape>
  <shape type="serialized" id="Shell1">
    <transform name="toWorld"><translate z="0.01"/><translate x="-2.5" y="1.5" z="-0.75"/></transform>
    <bsdf type="mybsdf"><integer name='idx' value='21'/><integer name='type' value='1'/></bsdf>
  </shape>
</scene>"""
    p = tmp_path / "scene.xml"
    p.write_text(xml)
    e = parse_matpreview_xml(str(p))
    assert e == [({"filename": "chm_orange_rgb"}, (-4.0, 1.0, 0.0)), ({"idx": 21, "type": 1}, (-2.5, 1.5, -0.75))]
    names, cam, centers, radii = scene_from_matpreview_xml(str(p), 64, 48)
    assert names == ["chm_orange_rgb", "bsdf_21"] and radii == [0.33, 0.33]
    assert centers == [(-4.0, 0.33, -1.0), (-2.5, 0.33, -1.5)] and (cam.width, cam.height) == (64, 48)
    (tmp_path / "empty.xml").write_text("<scene version='0.5.0'></scene>")
    with pytest.raises(ValueError, match="no <bsdf"):
        scene_from_matpreview_xml(str(tmp_path / "empty.xml"))
