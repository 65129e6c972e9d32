"""Plugin-level accuracy of the HIP path on EVERY shipped plugin weight set, as a statistic.

For each of the 27 disk + 25 spherical + 25 full-sphere ``bsdf_<i>`` sets, at N queries (default 65 536):

  * ``sample``  plugin-level ``sample()`` with an injected base point x0 -> (wo [N,3], pdf_sa [N])
                (rendering/brdf_measured_disk.py:59-82, brdf_measured_spherical.py:35-39,69-91, bsdf_myresult.py:59-84),
  * ``pdf_a``   plugin-level ``pdf()`` at the directions ``sample()`` produced (fp32, as a renderer hands them back),
  * ``pdf_b``   plugin-level ``pdf()`` at fresh directions, uniform over the (hemi)sphere
                (brdf_measured_disk.py:112-124, brdf_measured_spherical.py:122-137, bsdf_myresult.py:115-133),

through the C ABI on both tilings (32- and 16-query tiles), against the pinned fp64 oracle (oracle/bsdf_oracle.py) on the same
inputs — and, on the same rows, the oracle run in fp32 arithmetic exactly as the reference writes it (``acos(z/(r+1e-8))``
included): the reference's own distance from the fp64 answer.

Rows counted: ``det`` — the contract's error metric: density resolved, |p_ref| > 1e-6 x its 99th percentile,
and |prod det J| in [1e-3, 1e3]; ``all`` — every resolved row (what tests/test_gpu_parity.py's plugin-level fixtures count).
p99 carries a percentile-bootstrap 95 % interval (B resamples of the rows).  What round 6 measured with it (N = 65 536,
profiles/r06_plugin_parity_77sets.json): on the default tiling every set's upper interval end is below 1e-4 except bsdf_23's
pdf() at fresh directions, where the reference's own fp32 evaluation is 1.5e-3 and the kernel 5.6e-4 (EXEMPT, by name).  The
16-query tiling held one (set, call) AT the bound — cc_amber_citrine_rgb_disk pdf() at fresh directions, 1.05e-4 [1.02e-4, 1.08e-4],
reference fp32 3.1e-5 — until its cause was found (profiles/HISTORY.md, round 6 #19: the output layer's W_lo rows were fp16
subnormals, a fixed perturbation of the weights that move the state) and removed (csrc/bsdfd.hip, BSDFD_WO_LO_SCALE): 4.9e-5;
KNOWN_ABOVE_BOUND is empty.

This module is shared by tests/test_gpu_parity77.py (the assertion) and tools/plugin_parity_sweep.py (the committed record,
profiles/r06_plugin_parity_77sets.json).  The oracle half runs in worker PROCESSES that import numpy + oracle only.
"""
from __future__ import annotations

import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

KINDS = ("sample", "pdf_a", "pdf_b")
BOUND = 1e-4          # north_star: PDF rel-err <= 1e-4
BOOT = 400            # bootstrap resamples


def all_sets():
    """[(stem, domain, full_sphere)] of every shipped weight set a plugin loads (the 64-wide *_complex teachers are not)."""
    from bsdf_diffusion_sampling_amd import weights as W
    out = []
    for dom in ("disk", "spherical"):
        for stem in W.list_shipped(dom):
            if stem.endswith("_complex"):
                continue
            out.append((stem, dom, stem.startswith("bsdf_")))
    return out


def _load(stem, dom):
    from bsdf_diffusion_sampling_amd import weights as W
    return W.load(W.shipped_path(stem[: -len(dom) - 1], dom))


def make_inputs(stem, dom, full, n):
    """Deterministic per-set inputs (seed = crc32 of the set's name): wi3 [n,3] unit vectors, x0 [n,2] a base draw of the
    set's own base density (fp64 oracle, rounded to fp32), wl3 [n,3] fresh directions for pdf_b — all fp32."""
    from oracle import bsdf_oracle as O
    rng = np.random.default_rng(zlib.crc32(stem.encode()))
    orc = O.Oracle(_load(stem, dom))
    if dom == "disk":   # uniform on the disk of radius 0.95
        r, a = 0.95 * np.sqrt(rng.random(n)), 2 * np.pi * rng.random(n)
        w2 = np.stack([r * np.cos(a), r * np.sin(a)], 1)
        wi3 = np.concatenate([w2, np.sqrt(np.maximum(1 - (w2 ** 2).sum(1, keepdims=True), 0))], 1).astype(np.float32)
        x0 = orc.base_sample(wi3[:, :2].astype(np.float64), rng.standard_normal((n, 2)))
    else:               # config 3: theta_i ~ U(0, 1.5) (the full-sphere sets: U(0, 3)), phi_i ~ U(-pi, pi), handed over as unit vectors
        th, ph = (3.0 if full else 1.5) * rng.random(n), (2 * rng.random(n) - 1) * np.pi
        wi3 = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)], 1).astype(np.float32)
        cond = O.cart_to_spher(wi3.astype(np.float64))
        mu, kappa = orc.base_von_mises_params(cond)
        x0 = orc.base_sample(cond, rng.standard_normal(n), phi=rng.vonmises(mu, kappa))
    z = rng.random(n) * (2.0 if full else 1.0) - (1.0 if full else 0.0)
    ph = (2 * rng.random(n) - 1) * np.pi
    rr = np.sqrt(np.maximum(1 - z * z, 0))
    wl3 = np.stack([rr * np.cos(ph), rr * np.sin(ph), z], 1).astype(np.float32)
    return {"wi3": wi3, "x0": x0.astype(np.float32), "wl3": wl3}


def gpu_eval(stem, dom, full, inp, tiles=(32, 16), precision="split3"):
    """The HIP path through the C ABI (FlowSampler, ctypes/torch binding as configured), per tiling:
    {tile: {"wo": [n,3], "sample": pdf_sa, "pdf_a": ..., "pdf_b": ...}} + "wo_a" = the fp32 directions pdf_a is asked at
    (the first tiling's samples)."""
    import torch
    from bsdf_diffusion_sampling_amd import _lib
    from bsdf_diffusion_sampling_amd.sampler import FlowSampler
    fw = _load(stem, dom)
    T = 4 if dom == "disk" else 8
    variant = _lib.PLUGIN_FULLSPHERE if full else _lib.PLUGIN_MEASURED
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()  # noqa: E731
    wi, x0, wl = t(inp["wi3"]), t(inp["x0"]), t(inp["wl3"])
    out, wo_a = {}, None
    for tile in tiles:
        s = FlowSampler(fw, precision=precision, tile=tile if precision == "split3" else 0)
        assert s.tile == tile or precision != "split3", (stem, tile, s.tile)
        wo, pdf = s.plugin_sample(wi, x0, T=T, variant=variant)
        if wo_a is None:
            wo_a = wo.clone()
        pa = s.plugin_pdf(wi, wo_a, T=T, variant=variant)
        pb = s.plugin_pdf(wi, wl, T=T, variant=variant)
        out[tile] = {"wo": wo.cpu().numpy(), "sample": pdf.cpu().numpy(), "pdf_a": pa.cpu().numpy(), "pdf_b": pb.cpu().numpy()}
        s.close()
    out["wo_a"] = wo_a.cpu().numpy()
    return out


def oracle_eval(args):
    """(worker process) fp64 and fp32-arithmetic oracle on one set's inputs: {dtype: {kind: pdf, kind+"_acc": prod det, "wo": ...}}."""
    stem, dom, full, inp, wo_a = args
    from oracle import bsdf_oracle as O
    fw = _load(stem, dom)
    T = 4 if dom == "disk" else 8
    res = {}
    for name, dt in (("f64", np.float64), ("f32", np.float32)):
        orc = O.Oracle(fw, dt)
        wi, x0, wl, woa = (inp["wi3"].astype(dt), inp["x0"].astype(dt), inp["wl3"].astype(dt), wo_a.astype(dt))
        with np.errstate(all="ignore"):
            if dom == "disk":
                wo, ps, acc = O.plugin_sample_disk(orc, wi, x0, T=T, return_acc=True)
                pa, acca = O.plugin_pdf_disk(orc, wi, woa, T=T, return_acc=True)
                pb, accb = O.plugin_pdf_disk(orc, wi, wl, T=T, return_acc=True)
            else:
                wo, ps, acc = O.plugin_sample_spherical(orc, wi, x0, T=T, full_sphere=full, return_acc=True)
                pa, acca = O.plugin_pdf_spherical(orc, wi, woa, T=T, full_sphere=full, return_acc=True)
                pb, accb = O.plugin_pdf_spherical(orc, wi, wl, T=T, full_sphere=full, return_acc=True)
        res[name] = {"wo": wo, "sample": ps, "pdf_a": pa, "pdf_b": pb}
        if name == "f64":
            res[name].update(sample_acc=acc, pdf_a_acc=acca, pdf_b_acc=accb)
    return stem, res


def _p99_ci(e, rng, boot=BOOT):
    """p99 of e with a percentile-bootstrap 95 % interval."""
    n = e.shape[0]
    p99 = float(np.percentile(e, 99))
    if n < 200:
        return p99, p99, p99
    es = np.sort(e)
    # the p99 of a resample = an order statistic of the sorted sample at a binomially distributed rank: draw the ranks directly
    # (equivalent to resampling n rows with replacement and far cheaper than B x n gathers)
    k = int(np.ceil(0.99 * n)) - 1
    # number of resampled rows <= es[j] is Binomial(n, (j+1)/n); the resample's k-th order statistic is es[j] for the smallest
    # j with that count > k.  Invert through uniform order statistics: rank ~ the k-th order statistic of n uniforms, a Beta.
    u = rng.beta(k + 1, n - k, size=boot)
    idx = np.minimum((u * n).astype(np.int64), n - 1)
    b = es[idx]
    return p99, float(np.percentile(b, 2.5)), float(np.percentile(b, 97.5))


def stats(got, want, want32, acc, seed=0, exclude=None):
    """Error figures of one (set, kind, tiling).  Two row sets:

      * ``det`` — THE CONTRACT METRIC: density resolved (|p_ref| > 1e-6 x its 99th percentile) and
        |prod det J| in [1e-3, 1e3] (where a step's det J ~ 0 every fp32 evaluation, the reference's own included, loses digits);
      * ``all`` — every resolved row (what tests/test_gpu_parity.py's plugin-level fixtures count).

    For each: rows, median, p99 with its bootstrap 95 % interval, max — and the same for the reference's arithmetic (the oracle
    in fp32, lines as written) on the same rows.  Rows where the kernel AND the fp32-arithmetic reference both return exactly 0
    while the fp64 oracle does not are the reference's fp32 GUARD decisions (sin(theta_o) > 5e-5 on an fp32 acos,
    rendering/brdf_measured_spherical.py:133): the kernel reproduces them by design; counted, not scored."""
    rng = np.random.default_rng(seed)
    got = got.astype(np.float64)
    want32 = want32.astype(np.float64)
    finite = np.isfinite(want)
    scale = np.percentile(np.abs(want[finite]), 99)
    ok = finite & (np.abs(want) > 1e-6 * scale)
    guard = ok & (got == 0) & (want32 == 0)
    ok &= ~guard
    n_excl = 0
    if exclude is not None:      # disk plugins: rows within 1e-5 of the r^2 >= 0.995 guard whose decision differs (summarize())
        n_excl = int((ok & exclude).sum())
        ok &= ~exclude
    rel = lambda u: np.abs(u[ok] - want[ok]) / np.abs(want[ok])  # noqa: E731
    e, e32 = rel(got), rel(want32)
    okd = (np.abs(acc[ok]) > 1e-3) & (np.abs(acc[ok]) < 1e3)
    # a sign that differs from the fp64 oracle's is a failure — except on a row outside the det range where the reference's own
    # fp32 evaluation has no correct digit either (relative error > 1/2): a step's det J ~ 0 there and its sign is rounding noise
    # (seen once: 1 row of 262 144 of bsdf_5_spherical, prod det J = 1.3e9, fp64 2.0e8, reference fp32 4.5e6, kernel -1.4e8)
    sign = np.sign(got[ok]) != np.sign(want[ok])
    lost32 = ~okd & ~(e32 <= 0.5)
    out = {"nan": int((~np.isfinite(got)).sum()), "guard_rows_as_reference_fp32": int(guard.sum()), "threshold_rows": n_excl,
           "sign_mismatch": int((sign & ~lost32).sum()), "sign_mismatch_where_reference_fp32_lost": int((sign & lost32).sum()),
           "zero_mismatch": int((((got == 0) != (want == 0)) & ~guard & ((want == 0) | (np.abs(want) > 1e-30))).sum())}
    for name, sel in (("det", okd), ("all", np.ones_like(okd))):
        ee, ee32 = e[sel], e32[sel]
        ee32 = ee32[np.isfinite(ee32)]
        p99, lo, hi = _p99_ci(ee, rng)
        r99, rlo, rhi = _p99_ci(ee32, rng)
        out[name] = {"rows": int(sel.sum()), "median": float(np.median(ee)), "p99": p99, "p99_lo": lo, "p99_hi": hi, "max": float(ee.max()),
                     "ref32_median": float(np.median(ee32)), "ref32_p99": r99, "ref32_p99_lo": rlo, "ref32_p99_hi": rhi,
                     "ref32_max": float(ee32.max())}
    return out


def summarize(stem, dom, full, g, o, tiles=(32, 16)):
    """One set's record from gpu_eval's and oracle_eval's outputs."""
    row = {"domain": dom, "full_sphere": bool(full)}
    seed = zlib.crc32(stem.encode())
    for tile in tiles:
        r = {}
        # disk plugins zero a sample whose r^2 >= 0.995 (rendering/brdf_measured_disk.py:69-75): a row whose fp64 r^2 is within 1e-5 of
        # the threshold is decided by the last bits of ANY fp32 evaluation.  Rows where kernel and oracle decide differently AND the
        # undecided side's r^2 is that close are counted ("threshold_rows"), not scored — for sample() and the directions.
        thr = None
        if dom == "disk":
            gw, ow = g[tile]["wo"].astype(np.float64), o["f64"]["wo"]
            gz, oz = (gw[:, 0] == 0) & (gw[:, 1] == 0), (ow[:, 0] == 0) & (ow[:, 1] == 0)
            r2 = np.where(gz, ow[:, 0] ** 2 + ow[:, 1] ** 2, gw[:, 0] ** 2 + gw[:, 1] ** 2)
            thr = (gz != oz) & (np.abs(r2 - 0.995) < 1e-5)
        for kind in KINDS:
            r[kind] = stats(g[tile][kind], o["f64"][kind], o["f32"][kind], o["f64"][kind + "_acc"], seed, thr if kind == "sample" else None)
        ew = np.abs(g[tile]["wo"].astype(np.float64) - o["f64"]["wo"])
        ew32 = np.abs(o["f32"]["wo"].astype(np.float64) - o["f64"]["wo"])
        if thr is not None and thr.any():
            ew, ew32 = ew[~thr], ew32[~thr]
        r["wo"] = {"p99": float(np.percentile(ew, 99)), "max": float(ew.max()), "ref32_p99": float(np.percentile(ew32, 99)),
                   "ref32_max": float(ew32.max())}
        row[f"tile{tile}"] = r
    return row


ALL_ROWS_FACTOR = 3.0   # every resolved row, near-singular steps included: at most this x the reference's own fp32 p99 (or the bound)
# (set, tiling, call) -> cap on the upper end of the p99 interval: measured above 1e-4 where the reference's fp32 evaluation is not.
# EMPTY since the 16-query kernels store the output layer's lo rows scaled (csrc/bsdfd.hip, BSDFD_WO_LO_SCALE): the one entry,
# ("cc_amber_citrine_rgb_disk", 16, "pdf_b") at 1.05e-4 [1.02e-4, 1.08e-4], now reads 4.9e-5.
KNOWN_ABOVE_BOUND = {}


def verdict(row, tiles=(32, 16), stem=None):
    """(failures, exempt, known) of one set's record (``known``: the KNOWN_ABOVE_BOUND entries of ``stem`` that fired, under their cap).

    Contract metric (``det`` rows): a (tiling, kind) passes when the UPPER end of its p99 interval is <= 1e-4; it is EXEMPT —
    listed by name, not a failure — when the reference's own fp32 evaluation is above the bound on the same rows too (lower end
    of ITS interval > 1e-4) and the kernel is no worse than 1.25 x that; otherwise it fails.
    Every resolved row (``all``): p99 <= max(1e-4, ALL_ROWS_FACTOR x the reference's fp32 p99 on the same rows).
    Directions: p99 <= 1e-5 and max <= 1e-4, or no worse than the reference's own fp32 maximum where that is above 1e-4.
    No NaN; signs equal on every scored row."""
    fails, exempt, known = [], [], []
    for tile in tiles:
        for kind in KINDS:
            s = row[f"tile{tile}"][kind]
            if s["nan"] or s["sign_mismatch"]:
                fails.append((tile, kind, "nan/sign", s["nan"], s["sign_mismatch"]))
            d, a = s["det"], s["all"]
            cap = KNOWN_ABOVE_BOUND.get((stem, tile, kind))
            if d["p99_hi"] > BOUND:
                if d["ref32_p99_lo"] > BOUND and d["p99"] <= 1.25 * d["ref32_p99"]:
                    exempt.append((tile, kind, d["p99"], d["ref32_p99"]))
                elif cap is not None and d["p99_hi"] <= cap:
                    known.append((tile, kind, d["p99"], d["p99_hi"], d["ref32_p99"]))
                else:
                    fails.append((tile, kind, "det", d["p99"], d["p99_hi"], d["ref32_p99"]))
            if a["p99"] > max(BOUND if cap is None else cap, ALL_ROWS_FACTOR * a["ref32_p99"]):
                fails.append((tile, kind, "all", a["p99"], a["ref32_p99"]))
        w = row[f"tile{tile}"]["wo"]
        if w["p99"] > 1e-5 or w["max"] > 1e-4:
            if not (w["p99"] <= 1e-5 and w["ref32_max"] > 1e-4 and w["max"] <= 2.0 * w["ref32_max"]):
                fails.append((tile, "wo", w["p99"], w["max"], w["ref32_max"]))
    return fails, exempt, known


def run(n=65536, sets=None, tiles=(32, 16), workers=None, log=print, precision="split3"):
    """The whole sweep: GPU evaluation in this process, the oracle in ``workers`` spawned processes.  -> {"summary", "sets"}."""
    import multiprocessing as mp
    import time
    sets = all_sets() if sets is None else sets
    if workers is None:
        workers = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    os.environ.setdefault("OMP_NUM_THREADS", "1")       # inherited by the workers: one BLAS thread each
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    t0 = time.time()
    ctx = mp.get_context("spawn")
    rows, pending = {}, {}
    with ctx.Pool(workers) as pool:
        for stem, dom, full in sets:
            inp = make_inputs(stem, dom, full, n)
            g = gpu_eval(stem, dom, full, inp, tiles, precision)
            pending[stem] = (dom, full, g, pool.apply_async(oracle_eval, ((stem, dom, full, inp, g["wo_a"]),)))
        log(f"parity77: GPU half of {len(sets)} sets done in {time.time() - t0:.0f} s; waiting for the oracle ({workers} workers)")
        for stem, (dom, full, g, fut) in pending.items():
            _, o = fut.get()
            rows[stem] = summarize(stem, dom, full, g, o, tiles)
    fails, exempt, known = {}, {}, {}
    for stem, row in rows.items():
        f, e, k = verdict(row, tiles, stem)
        if f:
            fails[stem] = f
        if e:
            exempt[stem] = e
        if k:
            known[stem] = k
    def worst(kind, tile, which="det", skip=()):
        cand = [(stem, rows[stem][f"tile{tile}"][kind][which]) for stem in rows if stem not in skip]
        best = max(cand, key=lambda x: x[1]["p99_hi"])
        return {"set": best[0], **{k: best[1][k] for k in ("p99", "p99_lo", "p99_hi", "ref32_p99", "max", "rows")}}
    summ = {"queries_per_set": n, "sets": len(rows), "tiles": list(tiles), "bound": BOUND, "bootstrap_resamples": BOOT,
            "metric": "det = SURVEY.md §8(d) rows (density resolved, |prod det J| in [1e-3, 1e3]); all = every resolved row",
            "worst_det": {f"tile{t}": {k: worst(k, t) for k in KINDS} for t in tiles},
            "worst_det_not_exempt": {f"tile{t}": {k: worst(k, t, skip=tuple(exempt)) for k in KINDS} for t in tiles},
            "worst_all": {f"tile{t}": {k: worst(k, t, "all") for k in KINDS} for t in tiles},
            "worst_wo_max": max(((stem, tile, rows[stem][f"tile{tile}"]["wo"]["max"]) for stem in rows for tile in tiles), key=lambda x: x[2]),
            "median_of_p99_det": {k: float(np.median([rows[s][f"tile{t}"][k]["det"]["p99"] for s in rows for t in tiles])) for k in KINDS},
            "median_of_ref32_p99_det": {k: float(np.median([rows[s][f"tile{tiles[0]}"][k]["det"]["ref32_p99"] for s in rows])) for k in KINDS},
            "threshold_rows": int(sum(rows[s][f"tile{t}"]["sample"].get("threshold_rows", 0) for s in rows for t in tiles)),
            "guard_rows_as_reference_fp32": int(sum(rows[s][f"tile{tiles[0]}"][k]["guard_rows_as_reference_fp32"] for s in rows for k in KINDS)),
            "sign_mismatch_where_reference_fp32_lost": {s: c for s, c in ((s, int(sum(rows[s][f"tile{t}"][k]["sign_mismatch_where_reference_fp32_lost"]
                                                                                    for t in tiles for k in KINDS))) for s in rows) if c},
            "failures": fails, "exempt_reference_fp32_also_above_bound": exempt, "known_above_bound_under_their_cap": known,
            "seconds": round(time.time() - t0, 1)}
    return {"summary": summ, "sets": rows}
