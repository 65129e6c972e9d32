#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_cfg
mkdir -p $OUT
cd $REPO
python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -s -k "config or context or mixed or more_than_64 or plan_with" --durations=5 > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; grep -v "^plugin_\|^full_size" $OUT/pytest.log | tail -25
