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
# energy next to time (the kernels are power-limited: what an optimisation buys is what it saves in joules per query)
ew = [w for w in dict.fromkeys(w for w, _ in keys) if any(w in r and "joule_per_Mquery" in r[w] for r in rows)]
if ew:
    print("%-14s" % "J/Mquery (ratio) | W | in-kernel MHz | Mcycles per sample launch" + "".join("%34s" % w for w in ew))
    for t in tags:
        line = "%-14s" % t
        for w in ew:
            rr = [r[w] for r in rows if r["tag"] == t and w in r and r[w].get("joule_per_Mquery")]
            if rr:
                j = float(np.median([x["joule_per_Mquery"] for x in rr]))
                j0 = [x[w]["joule_per_Mquery"] for x in rows if x["tag"] == tags[0] and w in x and x[w].get("joule_per_Mquery")]
                watts = float(np.median([x["watts"] for x in rr]))
                mhz = [x.get("sample_mhz") or x.get("mhz") for x in rr if (x.get("sample_mhz") or x.get("mhz"))]
                cyc = [x.get("sample_Mcycles") or (x.get("ms", 0) * (x.get("mhz") or 0) * 1e-3) for x in rr]
                line += "%9.4f(%5.3f)%5.0fW%5.0f%8.3f" % (j, j / float(np.median(j0)) if j0 else float("nan"), watts,
                                                         float(np.median(mhz)) if mhz else 0, float(np.median(cyc)) if cyc else 0)
            else:
                line += "%34s" % "-"
        print(line)
