#!/bin/bash
# Round 3, batch 2: context kernel-time A/B + the corrected wave-specialisation ubench (+ PMC).
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_b2
mkdir -p $OUT
cd $REPO
python3 tools/ctx_ab.py 15 > $OUT/ctx_ab.jsonl 2> $OUT/ctx_ab.err
cat $OUT/ctx_ab.jsonl
bash tools/r03_spec.sh > $OUT/spec.log 2>&1
cat gpurun_out/r03_spec/spec2.txt
