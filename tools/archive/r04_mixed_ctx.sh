#!/bin/bash
# mixed_16Mi with and without the per-query context (2.3 GB per wavefront at this size)
mkdir -p gpurun_out/r04_ab
O=gpurun_out/r04_ab/mixed_ctx.txt
: > $O
for c in on off on off; do
  python bench.py --workload mixed_16Mi --no-secondary --no-cpu-baseline --context $c --steps 6 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('mixed_16Mi context $c: value %.1f Msamples/s  ms_per_step %.3f  passes %d  avg flow launch %.4f ms  clock %.0f MHz' % (d['value'], d['ms_per_step'], d['config']['passes_per_step'], r['avg_launch_ms'], r.get('shader_clock_mhz') or 0))" >> $O
done
cat $O
