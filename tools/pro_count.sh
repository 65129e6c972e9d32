#!/bin/bash
# gpurun -- bash tools/pro_count.sh : executed-instruction counts of prologue + epilogue (tools/pro_count.py), disk and spherical
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pro_count; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for dom in disk spherical; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/$dom -o pmc -- python3 $REPO/tools/pro_count.py $dom > $OUT/$dom.log 2>&1
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections
CASES = ["sample", "sample_ctx_read", "sample_x0", "pdf", "pdf_ctx_read"]
for dom in ("disk", "spherical"):
    f = glob.glob(f"gpurun_out/pro_count/{dom}/**/pmc_counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "flow_kernel" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(per)[1:]   # skip the warm launch
    tiles = (1 << 20) / 16
    for i, c in enumerate(CASES):
        a, b = per[ids[i]], per[ids[i + len(CASES)]]
        out = []
        for k in sorted(a):
            if k == "SQ_WAVES" or k not in b:
                continue
            loop = (b[k] - a[k]) / tiles
            out.append(f"{k[9:]} fixed {a[k] / tiles - loop:7.1f} loop {loop:6.1f}")
        print(f"{dom:9s} {c:16s} per tile: " + " | ".join(out))
PY
