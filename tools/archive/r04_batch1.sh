#!/bin/bash
# Round 4, GPU batch 1: diagnostics behind DESIGN.md §4 "Round 4" (run through gpurun from the repo root)
#   counters list, MFMA operand-source micro-benchmark, A/B of the read-pinning / prefetch variants, occupancy sweep
cd "$(dirname "$0")/.."
O=gpurun_out/r04; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
timeout 300 tools/ubench/mfma_src > $O/mfma_src.txt 2>&1
tools/ab_run.sh 3 "--only disk8,disk4,sph8" base pin pins pinpf nl > $O/ab1.txt 2>&1
for pad in 0 36864 61440; do
  for lib in tune tunebase; do
    BSDFD_LDS_PAD=$pad BSDFD_LIB_PATH=$PWD/build_ab/lib_$lib.so timeout 600 python3 tools/ab.py --tag ${lib}_pad$pad --only disk8,sph8 | tail -1 >> $O/occ.jsonl
  done
done
cat $O/mfma_src.txt; cat $O/ab1.txt; cat $O/occ.jsonl
