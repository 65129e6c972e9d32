#!/usr/bin/env python3
"""Condense a tools/ab_run.sh .jsonl: per variant, median over rounds of each kernel time, ratio to the first variant."""
import collections, json, sys
import numpy as np
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip().startswith("{")]
tags = list(dict.fromkeys(r["tag"] for r in rows))
keys = [("disk8", "sample_ms"), ("disk8", "pdf_ms"), ("disk4", "sample_ms"), ("disk4", "pdf_ms"), ("sph8", "sample_ms"),
        ("sph8", "pdf_ms"), ("cplx8", "sample_ms"), ("cplx8", "pdf_ms"), ("teacher", "ms"), ("fused4", "ms"), ("fusedsph8", "ms")]
keys = [wk for wk in keys if any(wk[0] in r for r in rows)]
med = collections.defaultdict(dict)
for t in tags:
    for w, k in keys:
        v = [r[w][k] for r in rows if r["tag"] == t and w in r]
        if v:
            med[t][(w, k)] = float(np.median(v))
print("%-14s" % "variant" + "".join("%16s" % (w + "." + k.replace("_ms", "")) for w, k in keys))
for t in tags:
    line = "%-14s" % t
    for wk in keys:
        if wk in med[t]:
            line += "%9.4f(%5.3f)" % (med[t][wk], med[t][wk] / med[tags[0]][wk]) if wk in med[tags[0]] else "%16.4f" % med[t][wk]
        else:
            line += "%16s" % "-"
    print(line)
for t in tags:
    acc = [r["acc"] for r in rows if r["tag"] == t and "acc" in r]
    if acc:
        print(t, "p99:", " ".join("%s=%.1e/%.1e" % (k.replace("aniso_miro_7_rgb", "miro").replace("_spherical", "_sph"), v["p99"], v["pdf_p99"]) for k, v in acc[0].items()),
              "nan", sum(v["nan"] for v in acc[0].values()))
    fz = [r["fusedsph8"]["vs_two_calls_p999"] for r in rows if r["tag"] == t and "fusedsph8" in r and "vs_two_calls_p999" in r["fusedsph8"]]
    if fz:
        print(t, "fused spherical vs two launches (max |dwo|, p99.9 rel pdf_wo, p99.9 rel pdf_wl):", " ".join("%.1e" % x for x in fz[0]))
