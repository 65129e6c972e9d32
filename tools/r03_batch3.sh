#!/bin/bash
# Round 3, batch 3: VGPR-bank / literal ubench + the plugin-level golden parity tests + T-scan of the loop cost
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_b3
mkdir -p $OUT
cd $REPO
tools/ubench/bank > $OUT/bank.txt 2>&1
cat $OUT/bank.txt
rm -f gpurun_out/plugin_parity.jsonl
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "plugin_level or context" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest.log
cp gpurun_out/plugin_parity.jsonl $OUT/ 2>/dev/null
