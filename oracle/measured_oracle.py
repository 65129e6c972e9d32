"""ORACLE (test infrastructure only): numpy fp64 restatement of the ground-truth evaluator the
reference's plugins delegate ``eval()`` to — Mitsuba 3's ``measured`` BSDF on an RGL tensor file
(rendering/brdf_measured_disk.py:36-42 builds it with ``mi.load_dict({'type': 'measured', 'filename':
'measuredbsdfs/<name>.bsdf'})``; ``eval`` is called at :96,107).

Mitsuba is a third-party dependency that is absent from /root/reference and from this image, so this
is a restatement of the PUBLISHED model (Dupuy & Jakob, "An Adaptive Parameterization for Efficient
Material Acquisition and Rendering", SIGGRAPH Asia 2018, and the Mitsuba 3 plugin of the same authors):

    f(wi, wo) cos(theta_o) = spec(sample; phi_i, theta_i) * D(u_m) / (4 sigma(u_i))
    u = (sqrt(2 theta / pi), (phi + pi) / 2 pi)            unit-square coordinates of a direction
    sample = VNDF^{-1}(u_m | phi_i, theta_i)               inverse of the VNDF importance-sampling warp
    theta = elevation(w) = 2 asin(|w - z| / 2),  m = normalize(wi + wo)

with every table a bilinearly interpolated grid, linearly interpolated over the incident-direction
parameters (and the colour channel for `rgb`), the VNDF warp being the exact inverse-CDF map of that
interpolated density (marginal in v, conditional in u).

PARITY UNPINNED against Mitsuba itself (cannot run here).  What pins it instead (tests/):
  * the tensor-file reader reproduces the field table of the shipped file byte for byte;
  * warp consistency: sample(invert(p)) == p, the warp's Jacobian equals the interpolated density;
  * energy: the directional albedo of every available material is in (0, 1];
  * the reference's own trained networks: their density is trained to be proportional to
    lum(f cos) / cos (disk parameterisation, learning_repo_cleanup/utils/mitsuba_brdf_scalar.py:85-88),
    so the correlation between this evaluator and the shipped nets must be high.
"""
from __future__ import annotations

import struct

import numpy as np

_DTYPES = {1: np.uint8, 2: np.int8, 3: np.uint16, 4: np.int16, 5: np.uint32, 6: np.int32, 7: np.uint64,
           8: np.int64, 9: np.float16, 10: np.float32, 11: np.float64}


def read_tensor_file(path: str) -> dict:
    """Mitsuba ``TensorFile``: 'tensor_file\\0', u8 major, u8 minor, u32 n_fields, then per field
    u16 name_len, name, u16 ndim, u8 dtype, u64 offset, u64 shape[ndim]; payloads at `offset`."""
    raw = open(path, "rb").read()
    if raw[:12] != b"tensor_file\x00":
        raise ValueError(f"{path}: not a tensor file")
    if tuple(raw[12:14]) != (1, 0):
        raise ValueError(f"{path}: unsupported tensor file version {tuple(raw[12:14])}")
    n = struct.unpack_from("<I", raw, 14)[0]
    pos, out = 18, {}
    for _ in range(n):
        nl = struct.unpack_from("<H", raw, pos)[0]; pos += 2
        name = raw[pos:pos + nl].decode(); pos += nl
        nd = struct.unpack_from("<H", raw, pos)[0]; pos += 2
        dt = raw[pos]; pos += 1
        off = struct.unpack_from("<Q", raw, pos)[0]; pos += 8
        shape = struct.unpack_from(f"<{nd}Q", raw, pos); pos += 8 * nd
        out[name] = np.frombuffer(raw, dtype=_DTYPES[dt], count=int(np.prod(shape)), offset=off).reshape(shape)
    return out


class Marginal2D:
    """Bilinearly interpolated grid `data[..., H, W]` over [0,1]^2 with leading parameter axes that
    are interpolated linearly at `param_values`; optional inverse-CDF warp (marginal over y, then
    conditional over x), each parameter slice normalised separately."""

    def __init__(self, data, param_values=(), normalize=True, sampling=True):
        data = np.asarray(data, dtype=np.float64)
        self.params = [np.asarray(p, dtype=np.float64) for p in param_values]
        assert data.ndim == len(self.params) + 2
        self.h, self.w = data.shape[-2:]
        self.sampling = sampling
        flat = data.reshape((-1, self.h, self.w)).copy()
        if sampling or normalize:
            # integrals in PATCH units: a bilinear patch integrates to the mean of its 4 corners
            cond = 0.5 * (flat[:, :, :-1] + flat[:, :, 1:])            # [S, H, W-1] per-row segment integrals
            cond_cdf = np.concatenate([np.zeros_like(cond[:, :, :1]), np.cumsum(cond, -1)], -1)   # [S, H, W]
            row = cond_cdf[:, :, -1]                                   # [S, H] row integrals
            marg = 0.5 * (row[:, :-1] + row[:, 1:])
            marg_cdf = np.concatenate([np.zeros_like(marg[:, :1]), np.cumsum(marg, -1)], -1)      # [S, H]
            total = marg_cdf[:, -1]
            if normalize:
                scale = 1.0 / total
                flat *= scale[:, None, None]
                cond_cdf *= scale[:, None, None]
                marg_cdf *= scale[:, None]
            self.cond_cdf = cond_cdf.reshape(data.shape)
            self.marg_cdf = marg_cdf.reshape(data.shape[:-1])
        self.data = flat.reshape(data.shape)
        # a normalised table is a density per unit area of [0,1]^2: patch units -> x (W-1)(H-1)
        self.density_scale = float((self.w - 1) * (self.h - 1)) if normalize else 1.0

    # -- parameter interpolation --------------------------------------------------------------------
    def _slices(self, param):
        """-> list of (weight [N], index tuple of [N] arrays) over the 2^D corner slices."""
        n = len(param[0]) if self.params else None
        combos = [(1.0, ())]
        for vals, p in zip(self.params, param):
            p = np.asarray(p, dtype=np.float64)
            if len(vals) == 1:
                combos = [(w, idx + (np.zeros(len(p), dtype=np.int64),)) for w, idx in combos]
                continue
            i = np.clip(np.searchsorted(vals, p, side="right") - 1, 0, len(vals) - 2)
            t = np.clip((p - vals[i]) / (vals[i + 1] - vals[i]), 0.0, 1.0)
            combos = [(w * (1 - t), idx + (i,)) for w, idx in combos] + [(w * t, idx + (i + 1,)) for w, idx in combos]
        return combos

    def _gather(self, table, param, *ij):
        acc = 0.0
        for w, idx in self._slices(param):
            acc = acc + w * table[idx + ij]
        return acc

    # -- queries ----------------------------------------------------------------------------------------
    def _patch(self, pos):
        x = np.asarray(pos[0], dtype=np.float64) * (self.w - 1)
        y = np.asarray(pos[1], dtype=np.float64) * (self.h - 1)
        ix = np.clip(np.floor(x).astype(np.int64), 0, self.w - 2)
        iy = np.clip(np.floor(y).astype(np.int64), 0, self.h - 2)
        return ix, iy, x - ix, y - iy

    def eval(self, pos, param=()):
        ix, iy, fx, fy = self._patch(pos)
        v00 = self._gather(self.data, param, iy, ix); v10 = self._gather(self.data, param, iy, ix + 1)
        v01 = self._gather(self.data, param, iy + 1, ix); v11 = self._gather(self.data, param, iy + 1, ix + 1)
        return ((1 - fy) * ((1 - fx) * v00 + fx * v10) + fy * ((1 - fx) * v01 + fx * v11)) * self.density_scale

    def invert(self, pos, param=()):
        """position in [0,1]^2 -> (uniform variates (u0, u1) that `sample` maps to it, density)."""
        ix, iy, fx, fy = self._patch(pos)
        v00 = self._gather(self.data, param, iy, ix); v10 = self._gather(self.data, param, iy, ix + 1)
        v01 = self._gather(self.data, param, iy + 1, ix); v11 = self._gather(self.data, param, iy + 1, ix + 1)
        pdf = (1 - fy) * ((1 - fx) * v00 + fx * v10) + fy * ((1 - fx) * v01 + fx * v11)
        # conditional in x at height y: the row density is the y-interpolation of the two vertex rows
        c0 = (1 - fy) * v00 + fy * v01
        c1 = (1 - fy) * v10 + fy * v11
        part = fx * (c0 + 0.5 * fx * (c1 - c0))
        cdf0 = self._gather(self.cond_cdf, param, iy, ix); cdf1 = self._gather(self.cond_cdf, param, iy + 1, ix)
        r0 = self._gather(self.cond_cdf, param, iy, np.full_like(ix, self.w - 1))
        r1 = self._gather(self.cond_cdf, param, iy + 1, np.full_like(ix, self.w - 1))
        row = (1 - fy) * r0 + fy * r1
        with np.errstate(divide="ignore", invalid="ignore"):
            u0 = np.where(row > 0, (part + (1 - fy) * cdf0 + fy * cdf1) / row, 0.0)
        u1 = fy * (r0 + 0.5 * fy * (r1 - r0)) + self._gather(self.marg_cdf, param, iy)
        return (u0, u1), pdf * self.density_scale

    def sample(self, u, param=()):
        """Inverse of `invert` (used by the tests only): uniform variates -> position, density."""
        u0, u1 = (np.asarray(a, dtype=np.float64) for a in u)
        n = len(u0)
        # row: last vertex row whose marginal CDF is <= u1 (parameter-interpolated CDF)
        marg = np.stack([self._gather(self.marg_cdf, param, np.full(n, j)) for j in range(self.h)], 1)
        iy = np.clip((marg <= u1[:, None]).sum(1) - 1, 0, self.h - 2)
        r = np.stack([self._gather(self.cond_cdf, param, np.full(n, j), np.full(n, self.w - 1)) for j in range(self.h)], 1)
        r0, r1 = r[np.arange(n), iy], r[np.arange(n), iy + 1]
        rem = u1 - marg[np.arange(n), iy]
        # solve fy (r0 + fy (r1 - r0) / 2) = rem
        a = r1 - r0
        with np.errstate(divide="ignore", invalid="ignore"):
            fy = np.where(np.abs(a) > 1e-14 * (np.abs(r0) + np.abs(r1)), (np.sqrt(np.maximum(r0 * r0 + 2 * a * rem, 0)) - r0) / a,
                          rem / np.where(r0 != 0, r0, 1))
        fy = np.clip(fy, 0, 1)
        row = (1 - fy) * r0 + fy * r1
        target = u0 * row
        cond = np.stack([(1 - fy) * self._gather(self.cond_cdf, param, iy, np.full(n, i))
                         + fy * self._gather(self.cond_cdf, param, iy + 1, np.full(n, i)) for i in range(self.w)], 1)
        ix = np.clip((cond <= target[:, None]).sum(1) - 1, 0, self.w - 2)
        rem = target - cond[np.arange(n), ix]
        v00 = self._gather(self.data, param, iy, ix); v10 = self._gather(self.data, param, iy, ix + 1)
        v01 = self._gather(self.data, param, iy + 1, ix); v11 = self._gather(self.data, param, iy + 1, ix + 1)
        c0, c1 = (1 - fy) * v00 + fy * v01, (1 - fy) * v10 + fy * v11
        a = c1 - c0
        with np.errstate(divide="ignore", invalid="ignore"):
            fx = np.where(np.abs(a) > 1e-14 * (np.abs(c0) + np.abs(c1)), (np.sqrt(np.maximum(c0 * c0 + 2 * a * rem, 0)) - c0) / a,
                          rem / np.where(c0 != 0, c0, 1))
        fx = np.clip(fx, 0, 1)
        pdf = ((1 - fx) * c0 + fx * c1) * self.density_scale
        return ((ix + fx) / (self.w - 1), (iy + fy) / (self.h - 1)), pdf


def elevation(d):
    """Numerically robust angle to +z: 2 asin(|d - z| / 2) (Mitsuba `elevation`)."""
    dist = np.sqrt(d[:, 0] ** 2 + d[:, 1] ** 2 + (d[:, 2] - 1.0) ** 2)
    return 2.0 * np.arcsin(np.clip(0.5 * dist, 0.0, 1.0))


def theta2u(theta):
    return np.sqrt(theta * (2.0 / np.pi))


def phi2u(phi):
    return (phi + np.pi) * (0.5 / np.pi)


class MeasuredBSDF:
    """RGL measured BSDF (rgb flavour) — evaluation only."""

    def __init__(self, path: str):
        t = read_tensor_file(path)
        for k in ("phi_i", "theta_i", "sigma", "ndf", "vndf", "rgb", "jacobian"):
            if k not in t:
                raise ValueError(f"{path}: field {k!r} missing (spectral files are not supported, use *_rgb.bsdf)")
        self.fields = t
        self.description = bytes(t["description"]).decode(errors="replace") if "description" in t else ""
        phi_i, theta_i = t["phi_i"].astype(np.float64), t["theta_i"].astype(np.float64)
        self.isotropic = len(phi_i) <= 2
        self.jacobian = bool(t["jacobian"][0])
        self.reduction = 0
        if not self.isotropic:
            self.reduction = int(np.rint(2 * np.pi / (phi_i[-1] - phi_i[0])))
            mid = 0.5 * (phi_i[0] + phi_i[-1])
            self.fold = (-1.0 if np.cos(mid) < 0 else 1.0, -1.0 if np.sin(mid) < 0 else 1.0)
        self.ndf = Marginal2D(t["ndf"], (), normalize=False, sampling=False)
        self.sigma = Marginal2D(t["sigma"], (), normalize=False, sampling=False)
        self.vndf = Marginal2D(t["vndf"], (phi_i, theta_i), normalize=True, sampling=True)
        self.spectra = Marginal2D(t["rgb"], (phi_i, theta_i, np.arange(3.0)), normalize=False, sampling=False)

    def eval(self, wi, wo):
        """wi, wo [N,3] unit vectors in the local frame -> f(wi, wo) cos(theta_o), [N,3] rgb."""
        wi, wo = np.asarray(wi, dtype=np.float64).copy(), np.asarray(wo, dtype=np.float64).copy()
        active = (wi[:, 2] > 0) & (wo[:, 2] > 0)
        if self.reduction >= 2:
            # only phi_i in a half-plane (reduction 2) or quadrant (reduction 4) is stored; Mitsuba folds with
            # mulsign_neg into y <= 0 (x <= 0), where its files keep phi_i — here the target quadrant is read
            # off the file's phi_i range (untested against a real anisotropic file: none ships with the reference)
            fy = wi[:, 1] * self.fold[1] < 0
            fx = (wi[:, 0] * self.fold[0] < 0) if self.reduction == 4 else fy
            for v in (wi, wo):
                v[:, 0] = np.where(fx, -v[:, 0], v[:, 0])
                v[:, 1] = np.where(fy, -v[:, 1], v[:, 1])
        wm = wi + wo
        wm /= np.maximum(np.linalg.norm(wm, axis=1, keepdims=True), 1e-30)
        theta_i, phi_i = elevation(wi), np.arctan2(wi[:, 1], wi[:, 0])
        theta_m, phi_m = elevation(wm), np.arctan2(wm[:, 1], wm[:, 0])
        # unit-square coordinates: x = elevation, y = azimuth (tables are stored [azimuth][elevation])
        u_wi = (theta2u(theta_i), phi2u(phi_i))
        um_y = phi2u(phi_m - phi_i) if self.isotropic else phi2u(phi_m)
        um_y = um_y - np.floor(um_y)
        u_wm = (theta2u(theta_m), um_y)
        param = (phi_i, theta_i)
        (s0, s1), _ = self.vndf.invert(u_wm, param)
        spec = np.stack([self.spectra.eval((s0, s1), param + (np.full(len(s0), float(c)),)) for c in range(3)], 1)
        if self.jacobian:
            with np.errstate(divide="ignore", invalid="ignore"):
                spec = spec * (self.ndf.eval(u_wm) / (4.0 * self.sigma.eval(u_wi)))[:, None]
        return np.where(active[:, None], spec, 0.0)
