#!/bin/bash
# round 6, GPU job 1: the 77-set plugin-level parity sweep (N = 65 536, both tilings) + a baseline bench line of the inherited build
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out/r06
timeout 1500 python3 tools/plugin_parity_sweep.py --n 65536 --out gpurun_out/r06/plugin_parity_77sets_base.json > gpurun_out/r06/parity77_base.log 2>&1
echo "sweep rc=$?"; tail -c 3000 gpurun_out/r06/parity77_base.log
timeout 600 python3 bench.py > gpurun_out/r06/bench_base.json 2> gpurun_out/r06/bench_base.err
echo "bench rc=$?"; cut -c1-600 gpurun_out/r06/bench_base.json
