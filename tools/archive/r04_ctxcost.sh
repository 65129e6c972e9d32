#!/bin/bash
# what the per-query context costs / saves in the bench's own launch pattern vs the isolated launches of tools/fixed_cost.py
mkdir -p gpurun_out/r04_ab
O=gpurun_out/r04_ab/ctx_cost.txt
: > $O
for c in on off on off; do
  python bench.py --no-secondary --no-cpu-baseline --context $c --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('context $c: value %.1f  avg launch %.4f ms  sample %.4f  pdf %.4f  clock %.0f MHz' % (d['value'], r['avg_launch_ms'], r.get('sample_launch_ms', float('nan')), r.get('pdf_launch_ms', float('nan')), r.get('shader_clock_mhz', 0)))" >> $O
done
python tools/fixed_cost.py disk >> $O 2>&1
cat $O
