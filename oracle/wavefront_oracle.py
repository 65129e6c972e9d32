"""ORACLE (test infrastructure only — never imported by the product): numpy restatement of the
wavefront harness kernels (bsdf_diffusion_sampling_amd/csrc/wavefront.hip).

What is restated and what pins it:
  * primary rays — pixel index -> film position + jitter -> pinhole ray, the procedure of the
    reference's rendering/utils/mitsuba_helper.py:59-127 (Mitsuba's sensor is a third-party
    dependency absent from /root/reference; a pinhole `perspective` sensor is restated here);
  * power-heuristic MIS weight — mitsuba_helper.py:130-137, a^2 / (a^2 + b^2), 0 where a <= 0;
  * Philox4x32-10 — Salmon et al. SC'11; pinned by the published known-answer vectors of the
    Random123 distribution (kat_vectors: zero / all-ones / pi-digits), tests/test_wavefront_cpu.py;
  * sphere intersection, Duff et al. orthonormal basis, lat-long bilinear lookup — the harness'
    own scene (the reference's scene lives in Mitsuba XML + a serialized mesh); parity here is
    HIP kernel vs this restatement on the same seeds.
Parity for this row is therefore "pinned" for Philox and the MIS weight and harness-defined for
the rest; the neural path inside it (sample()/pdf()) is pinned by tests/golden as before.
"""
from __future__ import annotations

import numpy as np

F = np.float32


def philox4x32(k0, k1, c0, c1, c2, c3, rounds: int = 10):
    """Vectorised Philox4x32-R; all arguments broadcastable uint32 arrays -> 4 uint32 arrays."""
    k0, k1, c0, c1, c2, c3 = (np.asarray(a, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
                              for a in np.broadcast_arrays(k0, k1, c0, c1, c2, c3))
    m0, m1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(rounds):
        p0, p1 = m0 * c0, m1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ k0
        n1 = p1 & mask
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ k1
        n3 = p0 & mask
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return tuple(a.astype(np.uint32) for a in (c0, c1, c2, c3))


def _u01_half_open(x):   # [0, 1)
    return (x >> np.uint32(8)).astype(F) * F(1.0 / 16777216.0)


def _u01_open(x):        # (0, 1]
    return ((x >> np.uint32(8)).astype(F) + F(1.0)) * F(1.0 / 16777216.0)


def onb(n):
    """Duff et al. 2017 orthonormal basis; n [N,3] unit -> s, t."""
    n = n.astype(F)
    sign = np.copysign(F(1.0), n[:, 2])
    a = F(-1.0) / (sign + n[:, 2])
    b = n[:, 0] * n[:, 1] * a
    s = np.stack([F(1.0) + sign * n[:, 0] * n[:, 0] * a, sign * b, -sign * n[:, 0]], 1)
    t = np.stack([b, sign + n[:, 1] * n[:, 1] * a, -n[:, 1]], 1)
    return s.astype(F), t.astype(F)


def primary(scene: dict, row_begin: int, row_end: int, spp: int, seed: int, pass_idx: int, with_material=False):
    """scene: dict(origin, right, up, forward, tan_half_fov, width, height, center, radius [, spheres =
    [(centre, radius), ...] replacing center/radius, plane = dict(y, c0, c1, scale)]).
    -> wi, wl, nrm, dir, each [N,3] fp32 (path order: row, col, sample) [, material ids [N] int64:
    ball index, n_balls for the floor (its reflectance in the wi slot), n_balls + 1 for a miss]."""
    w, h = scene["width"], scene["height"]
    rows = np.arange(row_begin, row_end, dtype=np.int64)
    row = np.repeat(rows, w * spp)
    col = np.tile(np.repeat(np.arange(w, dtype=np.int64), spp), len(rows))
    s = np.tile(np.arange(spp, dtype=np.int64), len(rows) * w)
    gp = ((row * w + col) * spp + s).astype(np.uint64)
    u = philox4x32(seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, gp & np.uint64(0xFFFFFFFF), gp >> np.uint64(32),
                   pass_idx & 0xFFFFFFFF, 0x57617665)
    jx, jy = _u01_half_open(u[0]), _u01_half_open(u[1])
    fx = (col.astype(F) + jx) / F(w)
    fy = (row.astype(F) + jy) / F(h)
    thf = F(scene["tan_half_fov"])
    sx = (F(2.0) * fx - F(1.0)) * thf
    sy = (F(1.0) - F(2.0) * fy) * thf * (F(h) / F(w))
    right, up, fwd, o = (np.asarray(scene[k], dtype=F) for k in ("right", "up", "forward", "origin"))
    d = fwd[None, :] + sx[:, None] * right[None, :] + sy[:, None] * up[None, :]
    d = (d / np.sqrt((d * d).sum(1, keepdims=True))).astype(F)
    spheres = scene.get("spheres") or [(scene["center"], scene["radius"])]
    n = len(d)
    t_best = np.full(n, F(3.0e38), dtype=F)
    hit_k = np.full(n, -1, dtype=np.int64)
    for k, (ck, rk) in enumerate(spheres):
        oc = (o - np.asarray(ck, dtype=F)).astype(F)
        radius = F(rk)
        b = (d * oc[None, :]).sum(1)
        perp = oc[None, :] - b[:, None] * d
        disc = radius * radius - (perp * perp).sum(1)
        t = -b - np.sqrt(np.maximum(disc, F(0.0)))
        better = (disc > 0) & (t > 0) & (t < t_best)
        t_best = np.where(better, t, t_best).astype(F)
        hit_k = np.where(better, k, hit_k)
    plane_hit = np.zeros(n, dtype=bool)
    plane = scene.get("plane")
    if plane is not None:
        with np.errstate(divide="ignore", invalid="ignore"):
            tp = ((F(plane["y"]) - o[1]) / d[:, 1]).astype(F)
        plane_hit = (d[:, 1] < 0) & (tp > 0) & (tp < t_best)
        t_best = np.where(plane_hit, tp, t_best).astype(F)
        hit_k = np.where(plane_hit, -1, hit_k)
    hit = hit_k >= 0
    cs = np.asarray([np.asarray(ck, dtype=F) for ck, _ in spheres], dtype=F)[np.maximum(hit_k, 0)]
    rs = np.asarray([F(rk) for _, rk in spheres], dtype=F)[np.maximum(hit_k, 0)]
    occ = (o[None, :] - cs).astype(F)
    nn = (occ + np.where(hit, t_best, F(0.0))[:, None] * d) / rs[:, None]   # (misses: any finite vector, masked below)
    nn = (nn / np.sqrt((nn * nn).sum(1, keepdims=True))).astype(F)
    nn = np.where(hit[:, None], nn, F(0.0)).astype(F)
    safe_n = np.where(hit[:, None], nn, np.array([0, 0, 1], dtype=F))
    fs, ft = onb(safe_n)
    wi = np.stack([-(d * fs).sum(1), -(d * ft).sum(1), -(d * safe_n).sum(1)], 1)
    wi = np.where(hit[:, None], wi, np.array([0, 0, 1], dtype=F)).astype(F)
    material = np.where(hit, hit_k, len(spheres) + 1)
    if plane is not None:
        h = o[None, :] + np.where(plane_hit, t_best, F(0.0))[:, None] * d
        cx = np.floor(h[:, 0] * F(plane["scale"])).astype(np.int64)
        cz = np.floor(h[:, 2] * F(plane["scale"])).astype(np.int64)
        refl = np.where(((cx + cz) & 1) == 1, F(plane["c1"]), F(plane["c0"])).astype(F)
        nn = np.where(plane_hit[:, None], np.array([0, 1, 0], dtype=F), nn).astype(F)
        wi = np.where(plane_hit[:, None], refl[:, None], wi).astype(F)
        material = np.where(plane_hit, len(spheres), material)
    u2, u3 = _u01_open(u[2]), _u01_half_open(u[3])
    r = np.sqrt(u2)
    ang = F(6.28318530717958647692) * u3
    wl = np.stack([r * np.cos(ang), r * np.sin(ang), np.sqrt(np.maximum(F(1.0) - u2, F(0.0)))], 1).astype(F)
    if with_material:
        return wi, wl, nrm_f32(nn), d, material.astype(np.int64)
    return wi, wl, nrm_f32(nn), d


def nrm_f32(a):
    return np.ascontiguousarray(a, dtype=F)


def env_lookup(env, d):
    """env [H,W,3] lat-long (y up), d [N,3] unit -> [N,3]; u = atan2(x,-z)/2pi wrapped, v = acos(y)/pi."""
    h, w = env.shape[:2]
    d = d.astype(F)
    uu = np.arctan2(d[:, 0], -d[:, 2]).astype(F) * F(0.15915494309189533577)
    uu = uu - np.floor(uu)
    vv = np.arccos(np.clip(d[:, 1], -1, 1)).astype(F) * F(0.31830988618379067154)
    x, y = uu * F(w) - F(0.5), vv * F(h) - F(0.5)
    xf, yf = np.floor(x), np.floor(y)
    ax, ay = (x - xf).astype(F), (y - yf).astype(F)
    x0, y0 = xf.astype(np.int64), yf.astype(np.int64)
    x1, y1 = x0 + 1, y0 + 1
    x0, x1 = x0 % w, x1 % w
    y0, y1 = np.clip(y0, 0, h - 1), np.clip(y1, 0, h - 1)
    w00, w10 = ((1 - ax) * (1 - ay))[:, None], (ax * (1 - ay))[:, None]
    w01, w11 = ((1 - ax) * ay)[:, None], (ax * ay)[:, None]
    return (w00 * env[y0, x0] + w10 * env[y0, x1] + w01 * env[y1, x0] + w11 * env[y1, x1]).astype(F)


def mis_power(pa, pb):
    """mitsuba_helper.py:130-137: select(pdf_a > 0, a^2 / (b^2 + a^2), 0)."""
    pa, pb = pa.astype(np.float64), pb.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        w = pa * pa / (pb * pb + pa * pa)
    return np.where(pa > 0, w, 0.0)


def shade(scene: dict, env, spp: int, wo, pdf_o, wl, pdf_l, nrm, dir_, f_o=None, f_l=None, wi=None, material=None):
    """-> per-pixel mean over spp of the one-bounce MIS estimate, [npix, 3] (fp64 accumulation).
    f_o / f_l: ground-truth f cos (albedo included) at (wi, wo) / (wi, wl); None -> proxy f cos = albedo pdf."""
    env = env.astype(np.float64)
    n = nrm.astype(np.float64)
    miss = (nrm == 0).all(1)
    safe_n = np.where(miss[:, None], np.array([0.0, 0.0, 1.0]), n)
    fs, ft = onb(safe_n.astype(F))
    fs, ft = fs.astype(np.float64), ft.astype(np.float64)
    to_world = lambda v: v[:, 0:1] * fs + v[:, 1:2] * ft + v[:, 2:3] * safe_n
    inv_pi = 1.0 / np.pi
    pb = np.where(np.isfinite(pdf_o) & (pdf_o > 0), pdf_o, 0.0).astype(np.float64)
    wb = mis_power(pb, np.maximum(wo[:, 2], 0.0).astype(np.float64) * inv_pi)
    albedo = np.asarray(scene["albedo"], dtype=np.float64)[None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        weight_b = albedo if f_o is None else np.where(pb[:, None] > 0, f_o.astype(np.float64) / pb[:, None], 0.0)
    if f_o is not None:   # a NaN entry = no ground truth for that path: proxy
        weight_b = np.where(np.isnan(f_o[:, :1]), albedo, weight_b)
    Lb = wb[:, None] * weight_b * env_lookup(env, to_world(wo.astype(np.float64)).astype(F))
    pl = wl[:, 2].astype(np.float64) * inv_pi
    pbl = np.where(np.isfinite(pdf_l) & (pdf_l > 0), pdf_l, 0.0).astype(np.float64)
    gt_l = np.zeros(len(pl), dtype=bool) if f_l is None else ~np.isnan(f_l[:, 0])
    ok = (pl > 0) & ((pbl > 0) | gt_l)
    with np.errstate(divide="ignore", invalid="ignore"):
        wl_w = np.where(ok, mis_power(pl, pbl) / pl, 0.0)
    weight_l = albedo * pbl[:, None] if f_l is None else np.where(gt_l[:, None], np.nan_to_num(f_l.astype(np.float64)), albedo * pbl[:, None])
    Ll = wl_w[:, None] * weight_l * env_lookup(env, to_world(wl.astype(np.float64)).astype(F))
    L = Lb + Ll
    L = np.where(miss[:, None], env_lookup(env, dir_), L)
    if material is not None:  # diffuse floor, cosine-sampled: reflectance (in the wi slot) x radiance
        n_balls = len(scene.get("spheres") or [0])
        floor = material == n_balls
        L = np.where(floor[:, None], wi[:, 0:1].astype(np.float64) * env_lookup(env, to_world(wl.astype(np.float64)).astype(F)), L)
    return L.reshape(-1, spp, 3).mean(1)
