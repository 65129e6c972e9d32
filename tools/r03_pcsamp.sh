#!/bin/bash
# Round 3: PC-sampling (beta) attempt on the disk T=8 kernel + the round-start bench line.  gpurun -- bash tools/r03_pcsamp.sh
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_pcs
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $REPO/bench.py --no-cpu-baseline > $OUT/bench_start.json 2> $OUT/bench_start.err
for method in stochastic host_trap; do
  unit=cycles; interval=65536
  [ $method = host_trap ] && unit=time && interval=1
  timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $interval \
      --output-format csv -d $OUT/pcs_$method -o pcs -- python3 $REPO/tools/pcsamp_run.py 10 disk8 > $OUT/pcs_$method.log 2>&1
  echo "$method rc=$?" >> $OUT/rc.txt
done
ls -la $OUT/pcs_* | head -30
tail -3 $OUT/pcs_*.log
