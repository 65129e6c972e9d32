#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_b4
mkdir -p $OUT
cd $REPO
python3 -m pytest tests/test_gpu_bench.py -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest.log
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; python3 -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source']['status'], d['roofline']['issue_bound'].get('frac_of_issue_bound'))
print(json.dumps(d['cpu_baseline'], indent=1))
print({k:(round(v['value'],1), round(v['frac'],4), round(v['avg_launch_ms'],4), v['shader_clock_mhz']) for k,v in d['secondary'].items()})
"
python3 tools/cpu_probe.py 20 > $OUT/cpu_probe.txt 2>&1
cat $OUT/cpu_probe.txt
