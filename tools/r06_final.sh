#!/bin/bash
# round 6: the full GPU suite + smoke, the judged bench line (default command), the 77-set parity record, the render timings — final build
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r06; mkdir -p $O
bash tools/r06_tests.sh
timeout 900 python3 bench.py > $O/bench_line.json 2> $O/bench_line.err
echo "bench rc=$?"; cut -c1-300 $O/bench_line.json
timeout 900 python3 tools/plugin_parity_sweep.py --out $O/plugin_parity_77sets_final.json > $O/parity_final.log 2>&1
echo "sweep rc=$?"
timeout 300 python3 tools/bench_render.py --passes 256 > $O/render_512.txt 2>&1; tail -3 $O/render_512.txt
timeout 300 python3 tools/render_array.py --passes 64 > $O/render_array.json 2>&1; tail -1 $O/render_array.json
