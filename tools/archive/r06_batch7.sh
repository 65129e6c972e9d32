#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r06; mkdir -p $O
bash tools/ab_run.sh 3 "--only cplx8" c16 c32shared c32vec > $O/ab_cplx.txt 2>&1
tail -12 $O/ab_cplx.txt
