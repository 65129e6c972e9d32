#!/usr/bin/env python3
"""Condense tools/mixed_pmc.sh's counter passes: per form (direct | gather) and kernel, dispatches and counter totals; HBM-side bytes
per wavefront = FETCH_SIZE x 2 (the gfx950 correction of MI355X_MICROARCH.md, HBM section: FETCH_SIZE tallies 128-B requests at 64 B)
+ WRITE_SIZE, both in KB.  A wavefront = 4 flow launches (sample + pdf for the disk and the spherical group)."""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    for k in ("flow_kernel32", "flow_kernel", "bucket_scatter", "bucket_scan", "bucket_count", "gather_wi", "scatter_results"):
        if k in name:
            return k
    return name.split("(")[0][:60]


def main():
    d = sys.argv[1]
    out = {}
    for mode in ("direct", "gather"):
        tot = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.defaultdict(lambda: collections.defaultdict(int))
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            for f in glob.glob(os.path.join(d, f"{mode}_{ctr}", "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] != ctr:
                        continue
                    k = short(r["Kernel_Name"])
                    tot[k][ctr] += float(r["Counter_Value"])
                    n[k][ctr] += 1
        flow = sum(v["FETCH_SIZE"] for k, v in n.items() if k.startswith("flow_kernel"))
        waves = max(flow / 4.0, 1.0)
        per = {}
        for k in sorted(tot):
            per[k] = {"dispatches_per_wavefront": n[k]["FETCH_SIZE"] / waves,
                      "fetch_MB_per_wavefront": tot[k]["FETCH_SIZE"] * 2 * 1024 / 1e6 / waves,
                      "write_MB_per_wavefront": tot[k]["WRITE_SIZE"] * 1024 / 1e6 / max(sum(v["WRITE_SIZE"] for kk, v in n.items() if kk.startswith("flow_kernel")) / 4.0, 1.0)}
        sel = [k for k in per if k.startswith(("flow_kernel", "bucket", "gather_wi", "scatter_results"))]
        out[mode] = {"wavefronts_counted": waves, "kernels": per,
                     "fetch_MB_per_wavefront": sum(per[k]["fetch_MB_per_wavefront"] for k in sel),
                     "write_MB_per_wavefront": sum(per[k]["write_MB_per_wavefront"] for k in sel)}
        out[mode]["hbm_side_MB_per_wavefront"] = out[mode]["fetch_MB_per_wavefront"] + out[mode]["write_MB_per_wavefront"]
    n_l = 1 << 24
    out["algorithmic_MB_per_wavefront"] = n_l * (28 + 28 + 8) / 1e6   # sample 28 B + pdf 28 B per lane + the material id
    if out["gather"]["hbm_side_MB_per_wavefront"]:
        out["direct_over_gather"] = out["direct"]["hbm_side_MB_per_wavefront"] / out["gather"]["hbm_side_MB_per_wavefront"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
