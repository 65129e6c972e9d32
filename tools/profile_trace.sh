#!/bin/bash
# Kernel-trace rocprofv3 pass of a bench.py workload + one GRBM counter pass (shader clock): tools/profile_trace.sh <tag> --workload <name>
set -u
TAG=${1:-r02t}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py --no-cpu-baseline --no-secondary $* > $OUT/trace.log 2>&1
grep -h "^{" $OUT/trace.log | tail -1 > $OUT/bench_line.json
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_GRBM_GUI_ACTIVE_GRBM_COUNT -o pmc -- python3 $REPO/bench.py --steps 10 --warmup 2 --passes-per-step 1 --no-cpu-baseline --no-secondary $* > $OUT/pmc_GRBM_GUI_ACTIVE_GRBM_COUNT.log 2>&1
head -6 $OUT/trace/trace_kernel_stats.csv
