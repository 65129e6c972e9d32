"""CPU checks of the statistic behind the 77-set parity test (tests/parity77.py): what is scored, what is counted and not scored,
what fails.  Synthetic arrays only — no GPU, no oracle evaluation."""
import numpy as np

import parity77 as P


def _base(n=20000, seed=1):
    rng = np.random.default_rng(seed)
    want = np.exp(rng.normal(0, 1, n))
    acc = np.exp(rng.normal(0, 0.5, n))
    got = (want * (1 + 2e-6 * rng.standard_normal(n))).astype(np.float32)
    want32 = (want * (1 + 2e-6 * rng.standard_normal(n))).astype(np.float32)
    return want, acc, got, want32


def test_p99_interval_brackets_the_point_estimate_and_shrinks_with_rows():
    rng = np.random.default_rng(0)
    e_small, e_big = np.abs(rng.standard_normal(4000)), np.abs(rng.standard_normal(64000))
    p, lo, hi = P._p99_ci(e_small, np.random.default_rng(1))
    P2, lo2, hi2 = P._p99_ci(e_big, np.random.default_rng(1))
    assert lo <= p <= hi and lo2 <= P2 <= hi2
    assert (hi2 - lo2) < 0.5 * (hi - lo)                      # 16x the rows: a quarter of the width, give or take
    assert abs(P2 - 2.5758) < 0.05                             # p99 of |N(0, 1)|


def test_stats_scores_det_rows_and_counts_guards_lost_signs_and_threshold_rows():
    want, acc, got, want32 = _base()
    s = P.stats(got, want, want32, acc)
    assert s["nan"] == 0 and s["sign_mismatch"] == 0 and s["guard_rows_as_reference_fp32"] == 0 and s["threshold_rows"] == 0
    assert s["det"]["rows"] <= s["all"]["rows"] and 1e-6 < s["det"]["p99"] < 1e-5 and s["det"]["p99_lo"] <= s["det"]["p99"] <= s["det"]["p99_hi"]
    # rows the reference's fp32 guard zeroes and the kernel zeroes too: counted, not scored
    g2, w2 = got.copy(), want32.copy()
    g2[:7] = 0
    w2[:7] = 0
    s2 = P.stats(g2, want, w2, acc)
    assert s2["guard_rows_as_reference_fp32"] == 7 and s2["sign_mismatch"] == 0 and s2["all"]["rows"] == s["all"]["rows"] - 7
    # ... zeroed by the kernel alone: a sign mismatch (a failure)
    g3 = got.copy()
    g3[:3] = 0
    assert P.stats(g3, want, want32, acc)["sign_mismatch"] == 3
    # a flipped sign on a row OUTSIDE the det range where the reference's fp32 has no correct digit either: counted apart
    a4, g4, w4 = acc.copy(), got.copy(), want32.copy()
    a4[11], g4[11], w4[11] = 1e9, -got[11], 0.02 * want32[11]
    s4 = P.stats(g4, want, w4, a4)
    assert s4["sign_mismatch"] == 0 and s4["sign_mismatch_where_reference_fp32_lost"] == 1
    # ... the same flip where the reference's fp32 is fine, or inside the det range: a failure
    w5 = want32.copy()
    assert P.stats(g4, want, w5, a4)["sign_mismatch"] == 1
    a6 = acc.copy()
    assert P.stats(g4, want, w4, a6)["sign_mismatch"] == 1
    # threshold rows handed in by summarize(): excluded and counted
    ex = np.zeros(want.shape[0], bool)
    ex[:3] = True
    s7 = P.stats(g3, want, want32, acc, exclude=ex)
    assert s7["sign_mismatch"] == 0 and s7["threshold_rows"] == 3


def _row(p99=2e-5, hi=2.2e-5, ref=1e-5, all_p99=None, **kw):
    d = {"rows": 1000, "median": 1e-6, "p99": p99, "p99_lo": 0.9 * p99, "p99_hi": hi, "max": 1e-3, "ref32_median": 1e-6, "ref32_p99": ref,
         "ref32_p99_lo": 0.9 * ref, "ref32_p99_hi": 1.1 * ref, "ref32_max": 1e-3}
    a = dict(d, p99=all_p99 if all_p99 is not None else p99)
    s = {"nan": 0, "sign_mismatch": 0, "det": d, "all": a}
    s.update(kw)
    return s


def _set(over=None):
    r = {f"tile{t}": dict({k: _row() for k in P.KINDS}, wo={"p99": 1e-7, "max": 1e-6, "ref32_p99": 1e-7, "ref32_max": 1e-6}) for t in (32, 16)}
    for (t, k), v in (over or {}).items():
        r[f"tile{t}"][k] = v
    return r


def test_verdict_pass_fail_exempt():
    assert P.verdict(_set(), stem="x") == ([], [], [])
    # upper interval end above the bound: a failure ...
    f, e, k = P.verdict(_set({(32, "pdf_b"): _row(p99=9.5e-5, hi=1.01e-4)}), stem="x")
    assert f and f[0][:3] == (32, "pdf_b", "det") and not e and not k
    # ... unless the reference's own fp32 is above the bound on the same rows and the kernel no worse than 1.25 x it: exempt, by name
    f, e, k = P.verdict(_set({(32, "pdf_b"): _row(p99=5e-4, hi=6e-4, ref=1.4e-3, all_p99=6e-4)}), stem="x")
    assert not f and e and e[0][:2] == (32, "pdf_b")
    f, e, k = P.verdict(_set({(32, "pdf_b"): _row(p99=2.5e-3, hi=2.6e-3, ref=1.4e-3)}), stem="x")
    assert f and not e
    # a NaN or a scored sign mismatch fails whatever the percentiles say; so does a direction off by more than 1e-4
    assert P.verdict(_set({(16, "sample"): _row(nan=1)}), stem="x")[0]
    assert P.verdict(_set({(16, "sample"): _row(sign_mismatch=1)}), stem="x")[0]
    bad = _set()
    bad["tile32"]["wo"] = {"p99": 1e-7, "max": 0.9, "ref32_p99": 1e-7, "ref32_max": 5e-6}
    assert P.verdict(bad, stem="x")[0]
    # every resolved row: at most ALL_ROWS_FACTOR x the reference's fp32 (or the bound)
    assert P.verdict(_set({(32, "sample"): _row(all_p99=4e-4, ref=1e-5)}), stem="x")[0]
    assert P.KNOWN_ABOVE_BOUND == {}


def test_summarize_sets_disk_threshold_flips_aside():
    n = 4096
    rng = np.random.default_rng(3)
    want = np.exp(rng.normal(0, 1, n))
    acc = np.ones(n)
    ang = rng.uniform(0, 2 * np.pi, n)
    rad = 0.9 * np.sqrt(rng.random(n))
    wo = np.stack([rad * np.cos(ang), rad * np.sin(ang), np.sqrt(1 - rad ** 2)], 1)
    o = {"f64": {"wo": wo.copy(), **{k: want.copy() for k in P.KINDS}, **{k + "_acc": acc for k in P.KINDS}},
         "f32": {"wo": wo.astype(np.float32), **{k: want.astype(np.float32) for k in P.KINDS}}}
    # row 5 sits on the guard: the oracle keeps it (r^2 = 0.995 - 2e-7), the kernel zeroes it
    r5 = np.sqrt(0.995 - 2e-7)
    o["f64"]["wo"][5] = [r5, 0.0, np.sqrt(1 - r5 * r5)]
    o["f32"]["wo"][5] = o["f64"]["wo"][5]
    g = {t: {"wo": o["f64"]["wo"].astype(np.float32), **{k: want.astype(np.float32) for k in P.KINDS}} for t in (32, 16)}
    for t in (32, 16):
        g[t]["wo"][5] = [0, 0, 1]
        g[t]["sample"] = g[t]["sample"].copy()
        g[t]["sample"][5] = 0
    row = P.summarize("some_disk", "disk", False, g, o)
    assert row["tile32"]["sample"]["threshold_rows"] == 1 and row["tile32"]["sample"]["sign_mismatch"] == 0
    assert row["tile32"]["wo"]["max"] < 1e-6
    assert P.verdict(row, stem="some_disk")[0] == []
    # the same zeroing far from the threshold is a failure
    o["f64"]["wo"][5] = [0.5, 0.0, np.sqrt(0.75)]
    o["f32"]["wo"][5] = o["f64"]["wo"][5]
    row = P.summarize("some_disk", "disk", False, g, o)
    assert row["tile32"]["sample"]["threshold_rows"] == 0 and P.verdict(row, stem="some_disk")[0]
