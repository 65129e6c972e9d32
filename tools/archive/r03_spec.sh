#!/bin/bash
# Round 3: wave-specialisation micro-benchmark (tools/ubench/spec2.hip) + a PMC pass of the same binary.
# Run on the GPU box: gpurun -- bash tools/r03_spec.sh
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_spec
mkdir -p $OUT
cd $REPO/tools/ubench
./spec2 2000000 > $OUT/spec2.txt 2>&1
./specialized > $OUT/specialized.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY \
   --output-format csv -d $OUT/pmc -o pmc -- $REPO/tools/ubench/spec2 2000000 > $OUT/pmc.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -o pmc -- $REPO/tools/ubench/spec2 2000000 > $OUT/pmc_grbm.log 2>&1
cat $OUT/spec2.txt
