#!/bin/bash
# Round 3: per-query context — parity tests + bench A/B (context on / off, interleaved).  gpurun -- bash tools/r03_ctx.sh
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_ctx
mkdir -p $OUT
cd $REPO
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "context or mixed or fused" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" > $OUT/rc.txt
for r in 1 2; do
  for c in on off; do
    python3 bench.py --no-cpu-baseline --context $c > $OUT/bench_${c}_$r.json 2> $OUT/bench_${c}_$r.err
  done
done
tail -5 $OUT/pytest.log
python3 - <<'PY'
import json, glob, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "r03_ctx")
for f in sorted(glob.glob(out + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    sec = d.get("secondary", {})
    print(os.path.basename(f), f"{d['value']:.1f}", f"frac {d['roofline']['frac']:.4f}", f"sample {d['roofline'].get('sample_launch_ms')} pdf {d['roofline'].get('pdf_launch_ms')}",
          {k: round(v["value"], 1) for k, v in sec.items()})
PY
