#!/bin/bash
# Round 4, GPU batch 2: split-fp16 prologue (A/B + accuracy on all 77 weight sets), meet-in-the-middle in the fused spherical
# kernel, theta-tangent pre-activations from LDS; PMC pass of the saturated micro-benchmark stream
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r04; mkdir -p $O
tools/ab_run.sh 3 "--only disk8,disk4,sph8,fused4,fusedsph8" r3 zt sp fs new > $O/ab2.txt 2>&1
for v in r3 new; do
  BSDFD_LIB_PATH=$PWD/build_ab/lib_$v.so timeout 900 python3 tools/acc_sweep.py --tag $v --out $O/acc_$v.json | tail -1 >> $O/acc_sweep.jsonl
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY \
   --output-format csv -d $O/pmc_sat -o pmc -- $O/../../tools/ubench/mfma_src pmc > $O/pmc_sat.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/pmc_sat_grbm -o pmc -- $O/../../tools/ubench/mfma_src pmc > $O/pmc_sat_grbm.log 2>&1
cd - > /dev/null
cat $O/ab2.txt $O/acc_sweep.jsonl; tail -3 $O/pmc_sat.log
