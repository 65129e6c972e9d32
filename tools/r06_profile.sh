#!/bin/bash
# Round 6: rocprofv3 evidence of every figure of the bench line — the judged workload and configs[2] with PMC passes, the other
# secondary workloads with a kernel trace + a GRBM (shader clock) pass.  gpurun -- bash tools/r06_profile.sh
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash tools/profile.sh r06 > gpurun_out/prof_r06.log 2>&1
bash tools/profile.sh r06_sph --workload spherical_16Mi_T8 > gpurun_out/prof_r06_sph.log 2>&1
for wl in disk_1Mi_T4 mixed_16Mi teacher_64x6_4Mi_T128 complex64_1Mi_T8; do
  bash tools/profile_trace.sh r06_$wl --workload $wl > gpurun_out/prof_r06_$wl.log 2>&1
done
ls gpurun_out/ | grep prof_r06
tail -2 gpurun_out/prof_r06/trace.log | cut -c1-300
