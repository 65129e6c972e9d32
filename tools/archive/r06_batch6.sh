#!/bin/bash
# round 6, GPU job 6: the 64 x 6 Jacobian kernel on 32-query tiles — parity, both tilings, timing against the 16-query kernel
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_tilings.py tests/test_gpu_architectures.py -x -q -k "complex or tilings or architect or context or ragged" > $O/t_k32c.log 2>&1
echo "tests rc=$?"; tail -6 $O/t_k32c.log
for r in 1 2 3; do
  timeout 300 python3 bench.py --workload complex64_1Mi_T8 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline > $O/cplx_t32_$r.json 2>> $O/cplx.err
  BSDFD_TILE=16 timeout 300 python3 bench.py --workload complex64_1Mi_T8 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline > $O/cplx_t16_$r.json 2>> $O/cplx.err
done
python3 - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06/cplx_t*_?.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f, round(d['value'],1), round(r['frac'],4), 'avg launch ms', round(r['avg_launch_ms'],4), 'MHz', round(r['shader_clock_mhz'] or 0), 'tile', d['config'].get('tile_queries'), 'J/Mq', r.get('joule_per_Mquery'), 'W', r.get('socket_power_w'))
    except Exception as e: print(f, 'ERR', e)
P
