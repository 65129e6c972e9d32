#!/bin/bash
# Round 3: rocprofv3 evidence of the bench command (kernel trace + PMC passes), both judged workloads, + the full GPU suite
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash tools/profile.sh r03 > gpurun_out/prof_r03.log 2>&1
bash tools/profile.sh r03_sph --workload spherical_16Mi_T8 > gpurun_out/prof_r03_sph.log 2>&1
ls gpurun_out/prof_r03 gpurun_out/prof_r03_sph | head -40
tail -2 gpurun_out/prof_r03/trace.log | cut -c1-300
