#!/bin/bash
# the exact-fp32-MFMA validation kernels (16-query tiles) on the 27 disk sets: where does the 16-query split3 kernel's 1.05e-4 come from?
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO; mkdir -p gpurun_out/r06
python3 tools/plugin_parity_sweep.py --only _disk --tiles 16 --precision f32 --out gpurun_out/r06/parity_disk_f32.json > gpurun_out/r06/parity_disk_f32.log 2>&1
tail -3 gpurun_out/r06/parity_disk_f32.log | cut -c1-300
python3 - <<'P'
import json
d=json.load(open("gpurun_out/r06/parity_disk_f32.json")); S=d["sets"]
for k in ("sample","pdf_a","pdf_b"):
    top=sorted(S.items(), key=lambda kv:-kv[1]["tile16"][k]["det"]["p99"])[:4]
    print(k,[(s[:26],"%.1e/%.1e"%(r["tile16"][k]["det"]["p99"],r["tile16"][k]["det"]["ref32_p99"])) for s,r in top])
P
