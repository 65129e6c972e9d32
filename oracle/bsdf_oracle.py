"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU (numpy) restatement of the reference's neural-BSDF sampler hot path, in the
closed form of SURVEY.md Appendix A: the two autograd ``backward()`` calls the
reference issues per Euler step are replaced by the mathematically identical
forward-mode propagation of two tangent vectors.  Every function cites the
reference file:line it restates.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product
(``bsdf_diffusion_sampling_amd``) fails loudly when its HIP library is missing
rather than falling back to anything here.

Pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md §4), so this oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF,
imported unmodified in the build container by ``tests/golden/make_golden.py``;
the resulting arrays are committed under ``tests/golden/`` and
``tests/test_oracle_golden.py`` checks this module against them
(fp64 oracle vs fp32 reference: <= 2e-5 typical, see the test for the bounds).

dtype: every function computes in the dtype of ``dtype=`` (np.float64 default:
the high-precision spec the HIP kernels are compared with; np.float32 gives the
reference's arithmetic class).
"""
from __future__ import annotations

import numpy as np

DOMAIN_DISK = 0
DOMAIN_SPHERICAL = 1

# torch.distributions.von_mises (torch 2.10.0) polynomial coefficients for
# log I0(kappa); call sites rendering/utils/model.py:305,314.
_I0_SMALL = [1.0, 3.5156229, 3.0899424, 1.2067492, 0.2659732, 0.0360768, 0.0045813]
_I0_LARGE = [0.39894228, 0.01328592, 0.00225319, -0.00157565, 0.00916281,
             -0.02057706, 0.02635537, -0.01647633, 0.00392377]


def positional_encoding(y, bands):
    """rendering/utils/model.py:9-57 ``positional_encoding_1`` (include_input,
    log_sampling): [y, sin(2^0 y), cos(2^0 y), ..., sin(2^(P-1) y), cos(2^(P-1) y)],
    each block holding both input dims."""
    out = [y]
    for b in range(bands):
        f = y.dtype.type(2.0 ** b)
        out.append(np.sin(y * f))
        out.append(np.cos(y * f))
    return np.concatenate(out, axis=-1)


def _sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


def _silu_and_grad(z):
    s = _sigmoid(z)
    return z * s, s * (1.0 + z * (1.0 - s))


def log_i0(kappa):
    """torch ``_log_modified_bessel_fn(x, order=0)`` (two polynomials split at 3.75)."""
    k = kappa
    y = (k / 3.75) ** 2
    small = np.zeros_like(k) + _I0_SMALL[-1]
    for c in _I0_SMALL[-2::-1]:
        small = small * y + c
    small = np.log(small)
    kk = np.maximum(k, k.dtype.type(1e-30))
    yl = 3.75 / kk
    large = np.zeros_like(k) + _I0_LARGE[-1]
    for c in _I0_LARGE[-2::-1]:
        large = large * yl + c
    large = kk - 0.5 * np.log(kk) + np.log(large)
    return np.where(k < 3.75, small, large)


def _softplus(x):
    """nn.Softplus(beta=1, threshold=20)."""
    return np.where(x > 20.0, x, np.log1p(np.exp(np.minimum(x, 20.0))))


class Oracle:
    """Holds one weight set (a ``FlowWeights``-shaped object: attributes domain,
    width, n_hidden, pe_bands, base_pe_bands, w_in, w_hidden, w_out, base_*)."""

    def __init__(self, fw, dtype=np.float64):
        self.dtype = np.dtype(dtype)
        c = lambda a: np.asarray(a, dtype=self.dtype)
        self.domain = int(fw.domain)
        self.state_dim = 2 if self.domain == DOMAIN_DISK else 3
        self.pe_bands = int(fw.pe_bands)
        self.base_pe_bands = int(fw.base_pe_bands)
        self.w_in = c(fw.w_in)
        self.w_hidden = [c(w) for w in fw.w_hidden]
        self.w_out = c(fw.w_out)
        self.base_w1, self.base_b1 = c(fw.base_w1), c(fw.base_b1)
        self.base_w2, self.base_b2 = c(fw.base_w2), c(fw.base_b2)

    # ---- velocity net + 2x2 Jacobian ------------------------------------
    def velocity_jacobian(self, x, alpha, pe_cond):
        """v(x, alpha | omega_i) and d v / d x for one Euler step.

        Disk: rendering/utils/model.py:479-501 (``NN_cond_pos_simpler``, 2nd def);
        spherical: model.py:422-446 (``NN_cond_pos``) / :449-477 (64-wide) with
        the ``[theta, sin phi, cos phi]`` input built at
        rendering/utils/mlp_brdf_sampling.py:119-121.  Jacobian = forward-mode
        equivalent of the two ``backward`` calls at mlp_brdf_sampling.py:31-41.
        Returns v [N,2], d0 = dv/dx_0 [N,2], d1 = dv/dx_1 [N,2]."""
        n = x.shape[0]
        one = np.ones((n, 1), self.dtype)
        zero = np.zeros((n, 1), self.dtype)
        if self.domain == DOMAIN_DISK:
            s = x
            t0 = np.concatenate([one, zero], 1)
            t1 = np.concatenate([zero, one], 1)
        else:
            sp, cp = np.sin(x[:, 1:2]), np.cos(x[:, 1:2])
            s = np.concatenate([x[:, 0:1], sp, cp], 1)
            t0 = np.concatenate([one, zero, zero], 1)
            t1 = np.concatenate([zero, cp, -sp], 1)
        sd = self.state_dim
        a = np.full((n, 1), alpha, self.dtype)
        h = np.concatenate([s, a, pe_cond], 1)
        w1 = self.w_in
        z = h @ w1.T
        zt0 = t0 @ w1[:, :sd].T
        zt1 = t1 @ w1[:, :sd].T
        h, g = _silu_and_grad(z)
        t0, t1 = zt0 * g, zt1 * g
        for w in self.w_hidden:
            z, zt0, zt1 = h @ w.T, t0 @ w.T, t1 @ w.T
            h, g = _silu_and_grad(z)
            t0, t1 = zt0 * g, zt1 * g
        wo = self.w_out
        return h @ wo.T, t0 @ wo.T, t1 @ wo.T

    def flow(self, x, cond, T, reverse):
        """T explicit Euler steps with determinant tracking.
        forward (sampling): mlp_brdf_sampling.py:26-47 / :116-138 — alpha=t/T,
        x += v/T, acc /= det(I + J/T).  reverse (pdf): :77-99 / :153-176 —
        alpha=1-t/T, x -= v/T, acc *= det(I - J/T).  det is NOT abs()'d.
        Returns (x_T, acc)."""
        x = np.array(x, dtype=self.dtype)
        cond = np.asarray(cond, dtype=self.dtype)
        pe = positional_encoding(cond, self.pe_bands)
        acc = np.ones(x.shape[0], self.dtype)
        c = self.dtype.type(-1.0 / T if reverse else 1.0 / T)
        for t in range(T):
            alpha = (1.0 - t / T) if reverse else (t / T)
            v, d0, d1 = self.velocity_jacobian(x, alpha, pe)
            # rows as the reference builds them (mlp_brdf_sampling.py:44-46):
            # J_1 = e0 + c*grad(v_0), J_2 = e1 + c*grad(v_1)
            j00 = 1.0 + c * d0[:, 0]
            j01 = c * d1[:, 0]
            j10 = c * d0[:, 1]
            j11 = 1.0 + c * d1[:, 1]
            det = j00 * j11 - j01 * j10
            x = x + c * v
            acc = acc * det if reverse else acc / det
        return x, acc

    # ---- conditional base density ---------------------------------------
    def base_forward(self, cond):
        """rendering/utils/model.py:383-386 / :290-293: PE_3 -> Linear+b -> SiLU -> Linear+b."""
        cond = np.asarray(cond, dtype=self.dtype)
        pe = positional_encoding(cond, self.base_pe_bands)
        z = pe @ self.base_w1.T + self.base_b1
        h = z * _sigmoid(z)
        return h @ self.base_w2.T + self.base_b2

    def base_sample(self, cond, eps, phi=None):
        """Deterministic part of ``D_base.sample``.  Disk (model.py:387-392):
        x0 = loc + eps*exp(log_scale), eps [N,2] ~ N(0,1).  Spherical
        (model.py:298-307): theta = loc + eps*(exp(log_scale)+1e-3), eps [N,1];
        phi ~ VonMises(mu, kappa) must be supplied (``phi`` [N])."""
        o = self.base_forward(cond)
        eps = np.asarray(eps, dtype=self.dtype)
        if self.domain == DOMAIN_DISK:
            return o[:, :2] + eps.reshape(-1, 2) * np.exp(o[:, 2:4])
        th = o[:, 0] + eps.reshape(-1) * (np.exp(o[:, 1]) + self.dtype.type(1e-3))
        return np.stack([th, np.asarray(phi, dtype=self.dtype)], 1)

    def base_von_mises_params(self, cond):
        """(mu, kappa) of the phi marginal, model.py:294-297: kappa = softplus(o3)+1e-3."""
        o = self.base_forward(cond)
        return o[:, 2], _softplus(o[:, 3]) + self.dtype.type(1e-3)

    def base_log_prob(self, x, cond):
        """Disk: model.py:393-398.  Spherical: model.py:308-317 (note ``- log_scale``
        while the residual is divided by ``exp(log_scale)+1e-3`` — kept as is)."""
        x = np.asarray(x, dtype=self.dtype)
        o = self.base_forward(cond)
        log2pi = self.dtype.type(np.log(2.0 * np.pi))
        if self.domain == DOMAIN_DISK:
            loc, ls = o[:, :2], o[:, 2:4]
            e = (x - loc) / np.exp(ls)
            return -log2pi - ls.sum(1) - 0.5 * (e * e).sum(1)
        loc, ls, mu = o[:, 0], o[:, 1], o[:, 2]
        kappa = _softplus(o[:, 3]) + self.dtype.type(1e-3)
        e = (x[:, 0] - loc) / (np.exp(ls) + self.dtype.type(1e-3))
        loggau = -0.5 * log2pi - ls - 0.5 * e * e
        logvon = kappa * np.cos(x[:, 1] - mu) - log2pi - log_i0(kappa)
        return loggau + logvon

    # ---- the four operators (rendering/utils/mlp_brdf_sampling.py) -------
    def network_sampling(self, omega_i, x0, T, return_acc=False):
        """``network_sampling_disk`` :17-51 (T=4) / ``network_sampling_spherical``
        :106-140 (T=8) with the base draw ``x0`` supplied by the caller (the
        reference's torch RNG stream is not reproducible elsewhere).
        Returns (x_T [N,2], pdf [N]); ``return_acc``: also prod 1/det J (the row
        filter of the error metric)."""
        p0 = np.exp(self.base_log_prob(x0, omega_i))
        x, acc = self.flow(x0, omega_i, T, reverse=False)
        return (x, p0 * acc, acc) if return_acc else (x, p0 * acc)

    def network_pdf(self, omega_o, omega_i, T, return_acc=False):
        """``network_pdf_disk`` :69-103 / ``network_pdf_spherical`` :144-181."""
        x, acc = self.flow(omega_o, omega_i, T, reverse=True)
        pdf = np.exp(self.base_log_prob(x, omega_i)) * acc
        return (pdf, acc) if return_acc else pdf


# ---------------------------------------------------------------------------
# Plugin-level tensor post-processing (the Mitsuba-free part of MyBSDF.sample/pdf)
# ---------------------------------------------------------------------------
def cart_to_spher(v):
    """rendering/brdf_measured_spherical.py:35-39."""
    r = np.sqrt((v * v).sum(1))
    theta = np.arccos(v[:, 2] / (r + v.dtype.type(1e-8)))
    phi = np.arctan2(v[:, 1], v[:, 0])
    return np.stack([theta, phi], 1)


def plugin_sample_disk(orc, wi3, x0, T=4, return_acc=False):
    """rendering/brdf_measured_disk.py:59-82 (everything before ``measured.eval``):
    omega_i = wi[:, :2]; flow; r^2 >= 0.995 -> wo=(0,0), pdf=0;
    z = sqrt(relu(1-r^2)) (rendering/utils/mitsuba_brdf_draw.py:40-43);
    pdf_sa = pdf * cos(theta_o).  The cos(theta_i) > 0 lane mask is applied by
    the caller to the *weight*, not to wo/pdf (:64,101), so it is not applied here."""
    wi3 = np.asarray(wi3, dtype=orc.dtype)
    wo2, pdf, acc = orc.network_sampling(wi3[:, :2], x0, T, return_acc=True)
    r2 = wo2[:, 0] ** 2 + wo2[:, 1] ** 2
    valid = r2 < 0.995
    wo2 = np.where(valid[:, None], wo2, 0.0)
    pdf = np.where(valid, pdf, 0.0)
    z = np.sqrt(np.maximum(1.0 - (wo2 ** 2).sum(1), 0.0))
    wo3 = np.concatenate([wo2, z[:, None]], 1)
    return (wo3, pdf * z, acc) if return_acc else (wo3, pdf * z)


def plugin_pdf_disk(orc, wi3, wo3, T=4, return_acc=False):
    """rendering/brdf_measured_disk.py:112-124."""
    wi3 = np.asarray(wi3, dtype=orc.dtype)
    wo3 = np.asarray(wo3, dtype=orc.dtype)
    pdf, acc = orc.network_pdf(wo3[:, :2], wi3[:, :2], T, return_acc=True)
    ok = (wi3[:, 2] > 0) & (wo3[:, 2] > 0)
    out = np.where(ok, pdf * wo3[:, 2], 0.0)
    return (out, acc) if return_acc else out


def frame_sin_theta(v):
    """Mitsuba 3 ``Frame3f.sin_theta(v)`` = sqrt(v.x^2 + v.y^2) (safe_sqrt of
    ``sin_theta_2``), used at brdf_measured_spherical.py:90,126."""
    return np.sqrt(v[:, 0] ** 2 + v[:, 1] ** 2)


def _inv_sin_clamped(sin_t, dtype):
    fmax = np.finfo(np.float32).max
    with np.errstate(divide="ignore"):
        return np.clip(1.0 / sin_t, 1.0, dtype.type(fmax))


def plugin_sample_spherical(orc, wi3, x0, T=8, full_sphere=False, return_acc=False):
    """rendering/brdf_measured_spherical.py:69-91 (``full_sphere=False``) and
    rendering/bsdf_myresult.py:59-84 (``full_sphere=True``: no cos(theta_o) guard,
    |sin theta_o| in the Jacobian)."""
    wi3 = np.asarray(wi3, dtype=orc.dtype)
    wo2, pdf, acc = orc.network_sampling(cart_to_spher(wi3), x0, T, return_acc=True)
    st, ct = np.sin(wo2[:, 0]), np.cos(wo2[:, 0])
    pdf = np.where(st > 0.00005, pdf, 0.0)
    if not full_sphere:
        pdf = np.where(ct > 0, pdf, 0.0)
    sp, cp = np.sin(wo2[:, 1]), np.cos(wo2[:, 1])
    wo3 = np.stack([cp * st, sp * st, ct], 1)  # sph_to_dir :31-34
    out = pdf * _inv_sin_clamped(frame_sin_theta(wo3), orc.dtype)
    return (wo3, out, acc) if return_acc else (wo3, out)


def plugin_pdf_spherical(orc, wi3, wo3, T=8, full_sphere=False, return_acc=False):
    """rendering/brdf_measured_spherical.py:122-137 / rendering/bsdf_myresult.py:115-133
    (the latter has neither the sin-theta guard nor the cos masks)."""
    wi3 = np.asarray(wi3, dtype=orc.dtype)
    wo3 = np.asarray(wo3, dtype=orc.dtype)
    wo2 = cart_to_spher(wo3)
    pdf, acc = orc.network_pdf(wo2, cart_to_spher(wi3), T, return_acc=True)
    inv = _inv_sin_clamped(frame_sin_theta(wo3), orc.dtype)
    if full_sphere:
        out = pdf * inv
    else:
        pdf = np.where(np.sin(wo2[:, 0]) > 0.00005, pdf, 0.0)
        ok = (wi3[:, 2] > 0) & (wo3[:, 2] > 0)
        out = np.where(ok, pdf * inv, 0.0)
    return (out, acc) if return_acc else out


# ---------------------------------------------------------------------------
# Config 1 (plumbing): 1-D toy flow, rendering/utils/model.py:78-98 ``NN``
# ---------------------------------------------------------------------------
def toy_flow_1d(params, x0, T=8):
    """cat[x, alpha] -> 4 x (Linear 64 + bias, SiLU) -> Linear 1 (+bias); T Euler
    steps with the scalar Jacobian 1 + (dv/dx)/T.
    ``params`` = list of (W [out,in], b [out]).  Returns (x_T, prod 1/(1+v'/T))."""
    x = np.asarray(x0, dtype=np.float64).reshape(-1, 1)
    acc = np.ones(x.shape[0])
    for t in range(T):
        a = np.full_like(x, t / T)
        h = np.concatenate([x, a], 1)
        th = np.concatenate([np.ones_like(x), np.zeros_like(x)], 1)
        for li, (w, b) in enumerate(params):
            z, zt = h @ w.T + b, th @ w.T
            if li < len(params) - 1:
                h, g = _silu_and_grad(z)
                th = zt * g
            else:
                h, th = z, zt
        x = x + h / T
        acc = acc / (1.0 + th[:, 0] / T)
    return x[:, 0], acc
