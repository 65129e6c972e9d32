#!/bin/bash
# round 6, GPU job 8: the compiler-only build (BSDFD_COMPILER_ONLY_BUILD=1) against the product build, then the profile passes of the final sources
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out/r06
timeout 1500 python3 -m pytest tests/test_gpu_fallback_build.py -x -q > gpurun_out/r06/t_fallback.log 2>&1
echo "fallback tests rc=$?"; tail -4 gpurun_out/r06/t_fallback.log
bash tools/r06_profile.sh
