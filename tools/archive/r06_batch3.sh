#!/bin/bash
# round 6, GPU job 3: RN split and two-product Jacobian A/B (time, J/query, accuracy on all 77 sets), direct vs gather by wavefront size,
# the bench line with per-workload energy
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r06; mkdir -p $O
bash tools/ab_run.sh 3 "--only disk8,disk4,sph8" base rn jac2x jac2w jac2xrn > $O/ab_rn_jac2.txt 2>&1
tail -22 $O/ab_rn_jac2.txt
for v in base rn jac2x jac2w jac2xrn; do
  BSDFD_LIB_PATH=$REPO/build_ab/lib_$v.so timeout 600 python3 tools/plugin_parity_sweep.py --tiles 32 --out $O/parity_$v.json > $O/parity_$v.log 2>&1
  echo "sweep $v rc=$?"
done
BSDFD_LIB_PATH=$REPO/build_ab/lib_rn16.so timeout 600 python3 tools/plugin_parity_sweep.py --tiles 16 --out $O/parity_rn16.json > $O/parity_rn16.log 2>&1
echo "sweep rn16 rc=$?"
python3 - <<'P'
import json
for v in ('base','rn','jac2x','jac2w','jac2xrn','rn16'):
    try:
        s=json.load(open(f'gpurun_out/r06/parity_{v}.json'))['summary']
        t=list(s['worst_det_not_exempt'])[0]
        print(v, {k:"%s %.2e [%.2e]"%(x['set'][:24],x['p99'],x['p99_hi']) for k,x in s['worst_det_not_exempt'][t].items()}, 'median', {k:"%.2e"%x for k,x in s['median_of_p99_det'].items()}, 'fails', len(s['failures']), 'exempt', list(s['exempt_reference_fp32_also_above_bound']))
    except Exception as e: print(v,'ERR',e)
P
timeout 900 python3 tools/mixed_direct_ab.py > $O/mixed_direct_ab.jsonl 2> $O/mixed_direct_ab.err
cat $O/mixed_direct_ab.jsonl | cut -c1-400
timeout 900 python3 bench.py --steps 10 --warmup 2 > $O/bench_energy.json 2> $O/bench_energy.err
echo "bench rc=$?"
python3 - <<'P'
import json
d=json.loads(open('gpurun_out/r06/bench_energy.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline'].get('board'))
for k,v in d['secondary'].items(): print(k, v.get('value'), v.get('frac'), v.get('joule_per_Mquery'), v.get('socket_power_w'), (v.get('board') or {}).get('source'), (v.get('board') or {}).get('samples'))
P
