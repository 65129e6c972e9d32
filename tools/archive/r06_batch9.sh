#!/bin/bash
# round 6, GPU job 9: the two kernel families of the FINAL build on one box (time, J/Mquery, clock, cycles), the 16-query kernels' PMC
# passes of the judged command, longer timed regions of the three main workloads
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r06; mkdir -p $O
bash tools/ab_run.sh 4 "--only disk8,disk4,sph8" final@16 final@32 > $O/ab_final_t16_vs_t32.txt 2>&1
tail -14 $O/ab_final_t16_vs_t32.txt
for wl in disk_1Mi_T8 spherical_16Mi_T8 mixed_16Mi; do
  steps=400; [ $wl != disk_1Mi_T8 ] && steps=200
  timeout 600 python3 bench.py --workload $wl --steps $steps --warmup 5 --no-secondary --no-cpu-baseline > $O/soak_$wl.json 2>> $O/soak.err
  python3 -c "
import json; d=json.loads(open('$O/soak_$wl.json').read().strip().splitlines()[-1]); print('$wl', round(d['value'],1), round(d['roofline']['frac'],4), 'timed', round(d['config']['timed_region_s'],2), 's', d['roofline'].get('joule_per_Mquery'))"
done
export BSDFD_TILE=16
bash tools/profile.sh r06_t16 > gpurun_out/prof_r06_t16.log 2>&1
unset BSDFD_TILE
ls gpurun_out/prof_r06_t16 | head -3
