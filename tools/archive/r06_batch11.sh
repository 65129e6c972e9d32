#!/bin/bash
# round 6, GPU job 11: layer 1's state operands with round-to-nearest hi parts on the 32-query kernels (BSDFD_T32_L1_RN): time, J/query,
# accuracy on all 77 sets (tile 32)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r06; mkdir -p $O
BSDFD_TILE=32 bash tools/ab_run.sh 4 "--only disk8,disk4,sph8" base l1rn > $O/ab_l1rn.txt 2>&1
tail -12 $O/ab_l1rn.txt
for v in base l1rn; do
  BSDFD_LIB_PATH=$REPO/build_ab/lib_$v.so timeout 900 python3 tools/plugin_parity_sweep.py --tiles 32 --out $O/parity_$v.json > $O/parity_$v.log 2>&1
  echo "sweep $v rc=$?"
done
python3 - <<'P'
import json
for v in ('base','l1rn'):
    try:
        s=json.load(open(f'gpurun_out/r06/parity_{v}.json'))['summary']
        print(v, {k:"%s %.2e [%.2e]"%(x['set'][:24],x['p99'],x['p99_hi']) for k,x in s['worst_det_not_exempt']['tile32'].items()}, 'median', {k:"%.2e"%x for k,x in s['median_of_p99_det'].items()}, 'fails', len(s['failures']), 'exempt', list(s['exempt_reference_fp32_also_above_bound']))
    except Exception as e: print(v,'ERR',e)
P
