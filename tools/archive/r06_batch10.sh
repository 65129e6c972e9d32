#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r06; mkdir -p $O
bash tools/ab_run.sh 3 "--only cplx8" final@16 final@32 c32w1@32 > $O/ab_cplx_w1.txt 2>&1
tail -10 $O/ab_cplx_w1.txt
