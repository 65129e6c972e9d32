"""Wavefront harness (BASELINE.json config 5: "matpreview render, 1024 spp").

The reference renders by handing its ``MyBSDF`` plugin to Mitsuba 3 and looping
``mi.render(scene, spp=4, seed)`` 128-256 times (rendering/brdf_measured_disk.py:146-155);
every pass the integrator calls the plugin's ``sample()`` once per path (and ``pdf()`` once
per emitter sample).  Mitsuba is not available to this build, so the harness supplies the
minimum around the SAME plugin entry points — primary rays as in
rendering/utils/mitsuba_helper.py:59-127, the power heuristic of :130-137, one analytic
sphere carrying the material, a lat-long environment map, one bounce — which is enough to
drive the hot path exactly the way a renderer does: one ``sample_t`` and one ``pdf_t`` call
per wavefront of (tile pixels x spp) paths per pass, image rows split across GPUs.

Per pass and tile (both streaming kernels live in csrc/wavefront.hip, C ABI ``bsdfd_wf_*``):

    primary  -> wi, wl (cosine light sample), normal, ray dir      48 B/path written
    sampler.plugin_sample_pdf(wi, wl) -> wo, pdf(wo), pdf(wl)      the hot path: sample() and pdf() of the
                                                                   same intersections in one launch
    shade    -> film tile += mean_spp of the MIS estimate          80 B/path read

The ground-truth ``eval()`` of the reference is Mitsuba's ``measured`` BSDF (not neural, not
part of this build).  The nets model pdf(wo | wi) ∝ lum(f cos), so the harness shades with
the proxy ``f cos = albedo * pdf`` — BSDF samples carry weight ``albedo``, light samples
``albedo * pdf_bsdf / pdf_light`` — a consistent one-bounce estimator whose cost profile
(one sample() + one pdf() per path) is that of the reference's render.

There is no CPU fallback: every step goes through libbsdfd.so.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import math
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from . import weights as W
from .sharding import shard_range


@dataclasses.dataclass
class Camera:
    """Pinhole camera; ``fov_deg`` is the horizontal field of view (Mitsuba's default fov_axis = x)."""
    origin: Sequence[float] = (0.0, 0.6, 3.2)
    target: Sequence[float] = (0.0, 0.0, 0.0)
    up: Sequence[float] = (0.0, 1.0, 0.0)
    fov_deg: float = 40.0
    width: int = 512
    height: int = 512

    def basis(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        o, t, u = (np.asarray(v, dtype=np.float64) for v in (self.origin, self.target, self.up))
        f = t - o
        f /= np.linalg.norm(f)
        r = np.cross(f, u)
        r /= np.linalg.norm(r)
        return r, np.cross(r, f), f


def make_sky(height: int = 256, width: int = 512, seed: int = 0) -> torch.Tensor:
    """Synthetic lat-long environment [H,W,3] (y up): horizon gradient + a sun + a few soft lights.
    (The reference's matpreview/envmap.exr needs an OpenEXR reader and is not redistributable data.)"""
    g = np.random.default_rng(seed)
    v = (np.arange(height) + 0.5) / height
    u = (np.arange(width) + 0.5) / width
    theta, phi = np.pi * v[:, None], 2 * np.pi * u[None, :]
    d = np.stack([np.sin(theta) * np.sin(phi), np.cos(theta) * np.ones_like(phi), -np.sin(theta) * np.cos(phi)], -1)
    up = np.clip(d[..., 1], 0, 1)[..., None]
    env = 0.25 * (1 - up) * np.array([1.0, 0.95, 0.9]) + 0.6 * up * np.array([0.45, 0.6, 1.0])
    env = np.where(d[..., 1:2] < 0, 0.08 * np.array([0.5, 0.45, 0.4]), env)
    for _ in range(5):
        c = g.normal(size=3)
        c[1] = abs(c[1]) + 0.2
        c /= np.linalg.norm(c)
        sharp = g.uniform(20, 400)
        col = g.uniform(0.5, 1.0, size=3) * g.uniform(2, 30)
        env = env + np.exp(sharp * (d @ c - 1.0))[..., None] * col
    return torch.from_numpy(env.astype(np.float32))


class WavefrontRenderer:
    """One material ball under an environment map, rendered in passes of ``spp`` samples per pixel.

    ``plugin`` is any of the ``MyBSDF`` mirrors (``sample_t`` / ``pdf_t``; its ``albedo`` tints the result).
    """

    def __init__(self, plugin, camera: Optional[Camera] = None, env: Optional[torch.Tensor] = None,
                 sphere_center: Sequence[float] = (0.0, 0.0, 0.0), sphere_radius: float = 1.0,
                 device: Optional[torch.device] = None, use_ground_truth: Optional[bool] = None):
        self.plugin = plugin
        # shade with the ground-truth f (the plugin's native `measured` evaluator) when it is available,
        # else with the proxy f cos = albedo * pdf_net
        has_gt = hasattr(getattr(plugin, "bsdf", None), "eval_t")
        self.use_ground_truth = has_gt if use_ground_truth is None else bool(use_ground_truth)
        if self.use_ground_truth and not has_gt:
            raise ValueError("use_ground_truth needs a plugin with a native MeasuredBSDF (props['measured_dir'])")
        self.needs_material_ids = False
        self.camera = camera or Camera()
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        env = make_sky() if env is None else env
        if env.dim() != 3 or env.shape[2] != 3:
            raise ValueError("env must be [H,W,3]")
        self.env = env.to(self.device, torch.float32).contiguous()
        r, u, f = self.camera.basis()
        sc = _lib.WfScene()
        for name, val in (("cam_origin", self.camera.origin), ("cam_right", r), ("cam_up", u), ("cam_forward", f),
                          ("sphere_center", sphere_center),
                          ("albedo", [float(a) for a in getattr(plugin, "albedo", torch.ones(3)).tolist()])):
            setattr(sc, name, (C.c_float * 3)(*[float(x) for x in val]))
        sc.tan_half_fov = math.tan(math.radians(self.camera.fov_deg) * 0.5)
        sc.width, sc.height = int(self.camera.width), int(self.camera.height)
        sc.sphere_radius = float(sphere_radius)
        sc.env_height, sc.env_width = int(self.env.shape[0]), int(self.env.shape[1])
        self.scene = sc
        self._buf = {}

    # -- buffers -----------------------------------------------------------------------------------
    def _buffers(self, n: int):
        b = self._buf.get(n)
        if b is None:
            mk = lambda *shape: torch.empty(shape, dtype=torch.float32, device=self.device)
            b = dict(wi=mk(n, 3), wl=mk(n, 3), nrm=mk(n, 3), dir=mk(n, 3), wo=mk(n, 3), pdf_o=mk(n), pdf_l=mk(n))
            if self.use_ground_truth:
                b.update(f_o=mk(n, 3), f_l=mk(n, 3))
            if self.needs_material_ids:
                b["mat"] = torch.empty((n,), dtype=torch.int64, device=self.device)
            self._buf = {n: b}  # one tile shape at a time
        return b

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # -- the two harness kernels ---------------------------------------------------------------------
    def primary(self, row_begin: int, row_end: int, spp: int, seed: int, pass_idx: int, out=None):
        """-> dict(wi, wl, nrm, dir), each [N,3], N = (row_end-row_begin) * width * spp."""
        n = (row_end - row_begin) * self.camera.width * spp
        b = out or self._buffers(n)
        p = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().bsdfd_wf_primary(C.byref(self.scene), row_begin, row_end, spp, seed, pass_idx,
                                                   p(b["wi"]), p(b["wl"]), p(b["nrm"]), p(b["dir"]),
                                                   p(b["mat"]) if "mat" in b else None, self._stream()))
        # written through raw pointers: tell torch (the plugin cores key their per-query context cache on wi._version)
        torch.autograd.graph.increment_version([b["wi"], b["wl"], b["nrm"], b["dir"]])
        return b

    def shade(self, row_begin: int, row_end: int, spp: int, b, film: torch.Tensor):
        """film [row_end-row_begin, width, 3] += the pass' estimate."""
        if film.shape != (row_end - row_begin, self.camera.width, 3) or film.dtype != torch.float32 \
                or not film.is_contiguous() or film.device != self.device:
            raise ValueError("film must be a contiguous fp32 [rows, width, 3] tensor on the renderer's device")
        p = lambda t: C.c_void_p(t.data_ptr())
        f_o = p(b["f_o"]) if "f_o" in b else None
        f_l = p(b["f_l"]) if "f_l" in b else None
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().bsdfd_wf_shade(C.byref(self.scene), p(self.env), row_begin, row_end, spp,
                                                 p(b["wo"]), p(b["pdf_o"]), p(b["wl"]), p(b["pdf_l"]), p(b["nrm"]),
                                                 p(b["dir"]), f_o, f_l, p(b["wi"]),
                                                 p(b["mat"]) if "mat" in b else None, p(film), self._stream()))

    # -- one pass over a tile ------------------------------------------------------------------------
    def render_pass(self, film: torch.Tensor, row_begin: int, row_end: int, spp: int, seed: int, pass_idx: int,
                    x0: Optional[torch.Tensor] = None):
        n = (row_end - row_begin) * self.camera.width * spp
        if n == 0:
            return
        b = self.primary(row_begin, row_end, spp, seed, pass_idx)
        core = self.plugin
        # sample(): Philox counter = global path index, key = (seed, pass) -> independent of the row split
        offset = row_begin * self.camera.width * spp
        skey = (seed * 0x9E3779B97F4A7C15 + pass_idx + 1) & 0xFFFFFFFFFFFFFFFF
        if core.DOMAIN == W.DOMAIN_DISK:
            # one launch for sample(wi) and pdf(wi, wl): the per-intersection prologue is shared in registers
            core.sampler.plugin_sample_pdf(b["wi"], b["wl"], x0, T=core.T, variant=core.VARIANT, seed=skey, offset=offset,
                                           out=(b["wo"], b["pdf_o"], b["pdf_l"]))
        else:
            # spherical nets: the single-op kernels form the Jacobian by meeting in the middle, the fused instantiation could not
            # (register budget) — two launches that share the prologue through the per-query context are the faster pair
            if "ctx" not in b:
                b["ctx"] = core.sampler.new_context(n)
            core.sampler.plugin_sample(b["wi"], x0, T=core.T, variant=core.VARIANT, seed=skey, offset=offset,
                                       out=(b["wo"], b["pdf_o"]), ctx_out=b["ctx"])
            core.sampler.plugin_pdf(b["wi"], b["wl"], T=core.T, variant=core.VARIANT, out=b["pdf_l"], ctx_in=b["ctx"])
        if self.use_ground_truth:  # eval() of the reference's loop: f cos (albedo-tinted) for both strategies
            core.bsdf.eval_t(b["wi"], b["wo"], out=b["f_o"], tint=core.albedo)
            core.bsdf.eval_t(b["wi"], b["wl"], out=b["f_l"], tint=core.albedo)
        self.shade(row_begin, row_end, spp, b, film)

    def render(self, passes: int, spp: int = 4, seed: int = 0, rows: Optional[Tuple[int, int]] = None) -> torch.Tensor:
        """Mean of ``passes`` passes of ``spp`` samples per pixel over rows [rows[0], rows[1]) (default:
        the whole film) -> [rows, width, 3].  The reference's loop: brdf_measured_disk.py:146-155
        (its `seed` is not advanced on the first iteration — fixed here: pass index = RNG stream)."""
        r0, r1 = rows if rows is not None else (0, self.camera.height)
        film = torch.zeros((r1 - r0, self.camera.width, 3), dtype=torch.float32, device=self.device)
        for k in range(passes):
            self.render_pass(film, r0, r1, spp, seed, k)
        return film / max(passes, 1)

    def render_sharded(self, passes: int, spp: int = 4, seed: int = 0, gather: bool = True):
        """Image-tile split (config 5): rank r renders a contiguous block of rows; the only exchange is
        the final gather of the film tiles (<= 3 MiB for 512^2) to rank 0.  Returns the full image on
        rank 0 (None elsewhere), or the local tile when ``gather`` is False."""
        import torch.distributed as dist
        from .sharding import gather_to_root
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        r0, r1 = shard_range(self.camera.height, rank, world)
        tile = self.render(passes, spp, seed, rows=(r0, r1))
        if world == 1 or not gather:
            return tile
        w = self.camera.width
        full = gather_to_root(tile.reshape(r1 - r0, w * 3), self.camera.height)
        return None if full is None else full.reshape(self.camera.height, w, 3)


# ball layout of the reference's matpreview/disney_bsdf_array0_envmap.xml (12 `mybsdf` balls in three rows;
# the scene is z-up with balls of radius ~0.5 translated by (x, y, z): here y-up, (x, y, z)_ref -> (x, z, -y))
ARRAY0_LAYOUT = [(-4.0, 1.0, 0.0), (-3.0, 2.0, 0.0), (-2.0, 3.0, 0.0), (-1.0, 4.0, 0.0),
                 (-3.5, 0.5, -0.75), (-2.5, 1.5, -0.75), (-1.5, 2.5, -0.75), (-0.5, 3.5, -0.75),
                 (-2.0, -0.5, -0.9), (-1.15, 0.35, -0.9), (-0.3, 1.2, -0.9), (0.55, 2.05, -0.9)]
ARRAY0_MATERIALS = ["aniso_brushed_aluminium_1_rgb", "aniso_copper_sheet_rgb", "aniso_green_pvc_rgb",
                    "aniso_metallic_paper_copper_rgb", "aniso_metallic_paper_gold_rgb", "aniso_miro_7_rgb",
                    "aniso_morpho_melenaus_rgb", "aniso_sari_silk_2color_rgb", "aurora_white_rgb",
                    "cc_amber_citrine_rgb", "cc_blue_agat_rgb", "cc_green_malachite_rgb"]


def array0_scene(width: int = 683, height: int = 512):
    """Camera, ball centres and radii approximating matpreview/disney_bsdf_array0_envmap.xml (sensor :362-381:
    origin (3.90, -3.46, 3.25) looking along (-0.65, 0.61, -0.44) in the z-up scene, fov 28.84 deg on the
    smaller axis, 1366x1024 film).  There the balls are the matpreview shell meshes on three steps of
    different height; here they are spheres of radius 0.33 resting on one floor, same (x, y) layout."""
    radius = 0.33
    centers = [(x, radius, -y) for x, y, _ in ARRAY0_LAYOUT]
    fov_smaller = 28.8415
    fov_x = 2 * math.degrees(math.atan(math.tan(math.radians(fov_smaller) / 2) * max(width / height, 1.0)))
    cx = sum(c[0] for c in centers) / len(centers)
    cz = sum(c[2] for c in centers) / len(centers)
    cam = Camera(origin=(3.89558, 3.25463 + 1.6, 3.46243), target=(cx, radius, cz), up=(0.0, 1.0, 0.0),  # raised: all 12 visible
                 fov_deg=fov_x, width=width, height=height)
    return cam, centers, [radius] * len(centers)


def parse_matpreview_xml(path: str):
    """Materials and ball positions of a reference scene file (rendering/matpreview/disney_bsdf_array*.xml):
    every <shape> that carries a <bsdf type="mybsdf"> contributes (props, (x, y, z)) — the string / integer
    properties of the bsdf (``filename`` or ``idx``) and the last <translate> of its toWorld transform that
    moves it in the ground plane.  Everything else in the file (meshes, integrator, emitter) belongs to Mitsuba."""
    import xml.etree.ElementTree as ET
    out = []
    for shape in ET.parse(path).getroot().iter("shape"):
        bsdf = next((b for b in shape.findall("bsdf") if b.get("type") == "mybsdf"), None)
        if bsdf is None:
            continue
        props = {}
        for child in bsdf:
            if child.tag in ("string", "integer", "float") and child.get("name") is not None:
                v = child.get("value")
                props[child.get("name")] = int(v) if child.tag == "integer" else float(v) if child.tag == "float" else v
        pos = (0.0, 0.0, 0.0)
        for tr in shape.iter("translate"):
            if tr.get("x") is not None or tr.get("y") is not None:
                pos = tuple(float(tr.get(a, 0.0)) for a in ("x", "y", "z"))
        out.append((props, pos))
    return out


def scene_from_matpreview_xml(path: str, width: int = 683, height: int = 512):
    """(material names, camera, centres, radii) for ``ArrayRenderer`` from a reference scene file; ball layout and
    camera direction as ``array0_scene`` (spheres of radius 0.33 on one floor, same ground-plane positions)."""
    entries = parse_matpreview_xml(path)
    if not entries:
        raise ValueError(f"{path}: no <bsdf type='mybsdf'> shapes")
    names = [str(p["filename"]) if "filename" in p else f"bsdf_{int(p['idx'])}" for p, _ in entries]
    radius = 0.33
    centers = [(x, radius, -y) for _, (x, y, _z) in entries]
    cam0, _, _ = array0_scene(width, height)
    cx = sum(c[0] for c in centers) / len(centers)
    cz = sum(c[2] for c in centers) / len(centers)
    cam = Camera(origin=cam0.origin, target=(cx, radius, cz), up=cam0.up, fov_deg=cam0.fov_deg, width=width, height=height)
    return names, cam, centers, [radius] * len(centers)


class ArrayRenderer(WavefrontRenderer):
    """Several material balls, one material each (``materials.MaterialTable``), over a checkerboard floor —
    the shape of the reference's matpreview array scenes (12 ``mybsdf`` instances per scene).  Per pass:
    primary rays (with material ids) -> one stable bucketing of the wavefront (``bsdfd_bucket_by_material``;
    floor hits and misses sort behind the materials) -> ONE fused sample+pdf launch per kernel signature
    (``bsdfd_plugin_sample_pdf_multi``) -> shade.  ``ground_truth``: {material index: MeasuredBSDF} for the
    balls that have an RGL tensor file; the others shade with the proxy ``f cos = albedo * pdf``.
    The Philox counter of a path is its row in the bucketed order, so images of different row splits agree
    statistically, not bit for bit (the single-ball renderer is split-invariant)."""

    def __init__(self, table, centers, radii, camera: Optional[Camera] = None, env: Optional[torch.Tensor] = None,
                 floor: bool = True, checker=(0.4, 0.2, 2.0), albedo=(1.0, 1.0, 1.0), ground_truth=None,
                 device: Optional[torch.device] = None):
        if len(centers) != len(table) or len(radii) != len(table):
            raise ValueError("one ball per material of the table")
        if not 1 <= len(table) <= 32:
            raise ValueError("1..32 balls")

        class _Tint:  # what WavefrontRenderer reads from a plugin
            pass
        tint = _Tint()
        tint.albedo = torch.tensor(albedo, dtype=torch.float32)
        tint.bsdf = None
        super().__init__(tint, camera, env, sphere_center=centers[0], sphere_radius=radii[0], device=device,
                         use_ground_truth=False)
        self.table = table
        self.ground_truth = dict(ground_truth or {})
        self.use_ground_truth = bool(self.ground_truth)   # f arrays are then filled for every path
        self.needs_material_ids = True
        sc = self.scene
        sc.n_extra_spheres = len(centers) - 1
        for k in range(1, len(centers)):
            sc.extra_spheres[k - 1] = (C.c_float * 4)(*[float(v) for v in centers[k]], float(radii[k]))
        sc.has_plane = 1 if floor else 0
        sc.plane_y = 0.0
        sc.checker_color0, sc.checker_color1, sc.checker_scale = (float(v) for v in checker)

    def render_pass(self, film: torch.Tensor, row_begin: int, row_end: int, spp: int, seed: int, pass_idx: int,
                    x0: Optional[torch.Tensor] = None):
        n = (row_end - row_begin) * self.camera.width * spp
        if n == 0:
            return
        b = self.primary(row_begin, row_end, spp, seed, pass_idx)
        plan = self.table.bucket(b["mat"], extra_bins=2)          # floor hits and misses behind the materials
        offset = row_begin * self.camera.width * spp
        skey = (seed * 0x9E3779B97F4A7C15 + pass_idx + 1) & 0xFFFFFFFFFFFFFFFF
        if not self.use_ground_truth:
            # (through the bucket permutation — no gathered copies of wi / wl, no scatter of the three results — while the pass's
            #  arrays fit the Infinity Cache: materials.WavefrontPipeline.DIRECT_MAX_LANES; the same numbers bit for bit)
            from .materials import WavefrontPipeline
            b["wo"], b["pdf_o"], b["pdf_l"] = self.table.sample_pdf(plan, b["wi"], b["wl"], seed=skey, offset=offset,
                                                                    direct=n <= WavefrontPipeline.DIRECT_MAX_LANES)
        else:
            # ground truth where a tensor file exists, evaluated on the bucket-ordered arrays (a material's rows
            # are contiguous there); NaN = "no ground truth for this path" -> the shade kernel uses the proxy
            b["wo"], b["pdf_o"], b["pdf_l"], s = self.table.sample_pdf(plan, b["wi"], b["wl"], seed=skey,
                                                                       offset=offset, return_bucketed=True)
            alb = self.plugin.albedo
            n_mat = s["wi"].shape[0]
            fo_s = torch.full((n_mat, 3), float("nan"), dtype=torch.float32, device=self.device)
            fl_s = torch.full((n_mat, 3), float("nan"), dtype=torch.float32, device=self.device)
            lo = 0
            for m, hi in enumerate(s["seg_end"]):
                if hi > lo and m in self.ground_truth:
                    self.ground_truth[m].eval_t(s["wi"][lo:hi], s["wo"][lo:hi], out=fo_s[lo:hi], tint=alb)
                    self.ground_truth[m].eval_t(s["wi"][lo:hi], s["wl"][lo:hi], out=fl_s[lo:hi], tint=alb)
                lo = hi
            b["f_o"].fill_(float("nan"))
            b["f_l"].fill_(float("nan"))
            b["f_o"][s["rows"]] = fo_s
            b["f_l"][s["rows"]] = fl_s
        self.shade(row_begin, row_end, spp, b, film)
