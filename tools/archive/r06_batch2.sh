#!/bin/bash
# round 6, GPU job 2: row_index (new tests + mixed_16Mi A/B), the acos diagnostic, the SPLIT_PRO=0 check of the 16-query disk kernels
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r06; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_row_index.py tests/test_gpu_configs.py::test_wavefront_pipeline_equals_the_serial_stages -x -q > $O/t_row_index.log 2>&1
echo "row_index tests rc=$?"; tail -5 $O/t_row_index.log
timeout 600 python3 tools/acos_diag.py --out $O/acos_diag.json > $O/acos_diag.txt 2>&1
echo "acos_diag rc=$?"; cat $O/acos_diag.txt
for lib in product nosplitpro; do
  ( [ $lib != product ] && export BSDFD_LIB_PATH=$REPO/build_ab/lib_$lib.so
    timeout 600 python3 tools/plugin_parity_sweep.py --n 65536 --only _disk --out $O/parity_disk_$lib.json > $O/parity_disk_$lib.log 2>&1
    echo "disk sweep $lib rc=$?" )
done
for r in 1 2 3; do
  timeout 300 python3 bench.py --workload mixed_16Mi --steps 10 --warmup 3 > $O/mixed_direct_$r.json 2>> $O/mixed.err
  BSDFD_BENCH_MIXED_GATHER=1 timeout 300 python3 bench.py --workload mixed_16Mi --steps 10 --warmup 3 > $O/mixed_gather_$r.json 2>> $O/mixed.err
done
python3 - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06/mixed_*_?.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],1), d['roofline'].get('frac'), d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
P
timeout 1200 python3 -m pytest tests/test_gpu_parity77.py tests/test_gpu_parity.py -x -q > $O/t_parity.log 2>&1
echo "parity tests rc=$?"; tail -5 $O/t_parity.log
cp gpurun_out/plugin_parity_77sets.json $O/plugin_parity_77sets_test.json 2>/dev/null
