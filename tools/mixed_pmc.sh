#!/bin/bash
# config 4 at 16 Mi lanes: HBM-side bytes per wavefront (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass, every kernel of
# the wavefront: bucketing, gather / scatter where they exist, the four flow launches) of the two forms of MaterialTable's pipeline:
#   direct  — the flow kernels read wi / write (wo, pdf, pdf) in lane order through the bucket permutation (bsdfd_opts.row_index)
#   gather  — round 5: gather of wi + scatter of the results as kernels of their own, bucket-ordered flow launches
# gpurun -- bash tools/mixed_pmc.sh   ->  gpurun_out/r06/mixed_pmc.json (+ the raw csv directories)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r06/mixed_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--workload mixed_16Mi --steps 3 --warmup 1 --passes-per-step 1 --no-cpu-baseline --no-secondary"
for mode in direct gather; do
  if [ $mode = gather ]; then export BSDFD_BENCH_MIXED_GATHER=1; else unset BSDFD_BENCH_MIXED_GATHER; fi
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d $O/${mode}_$ctr -o pmc -- python3 $REPO/bench.py $ARGS > $O/${mode}_$ctr.log 2>&1
  done
done
unset BSDFD_BENCH_MIXED_GATHER
python3 $REPO/tools/mixed_pmc_summary.py $O > $REPO/gpurun_out/r06/mixed_pmc.json
cat $REPO/gpurun_out/r06/mixed_pmc.json | head -60
