#!/bin/bash
# round 6, GPU job 4: HBM-side bytes of config 4's two forms (PMC), their steady-state timing by wavefront size, J/query of the
# two-product-Jacobian negative (the energy probe now reads the hwmon of THIS GPU)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r06; mkdir -p $O
bash tools/mixed_pmc.sh > $O/mixed_pmc.log 2>&1; tail -45 $O/mixed_pmc.log
cd $REPO
timeout 900 python3 tools/mixed_direct_ab.py --sizes 18,20,21,22,23,24 > $O/mixed_direct_ab2.jsonl 2> $O/mixed_direct_ab2.err
python3 - <<'P'
import json
for l in open('gpurun_out/r06/mixed_direct_ab2.jsonl'):
    d=json.loads(l); print(d['lanes'], 'direct %.3f ms (kernels %.3f)  gather %.3f ms (kernels %.3f)  ratio %.4f'%(d['direct']['wall_ms'],d['direct']['flow_kernel_ms'],d['gather']['wall_ms'],d['gather']['flow_kernel_ms'],d['direct_over_gather_wall']))
P
bash tools/ab_run.sh 3 "--only disk8,sph8" base jac2x rn > $O/ab_jac2_energy.txt 2>&1
tail -12 $O/ab_jac2_energy.txt
