#!/bin/bash
# round 6, GPU job 5: the priced spherical items (split-fp16 conditioning term with four products; d/dtheta tangent through a folded
# matrix) — time, J/query, accuracy on the 50 spherical-domain sets; the renderer / pipeline tests with the row-index path; array render
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r06; mkdir -p $O
bash tools/ab_run.sh 4 "--only sph8" base sphcond sphfold sphboth > $O/ab_sph_items.txt 2>&1
tail -16 $O/ab_sph_items.txt
for v in base sphcond sphfold sphboth; do
  BSDFD_LIB_PATH=$REPO/build_ab/lib_$v.so timeout 600 python3 tools/plugin_parity_sweep.py --tiles 32 --only _spherical --out $O/parity_sph_$v.json > $O/parity_sph_$v.log 2>&1
  echo "sweep $v rc=$?"
done
python3 - <<'P'
import json
for v in ('base','sphcond','sphfold','sphboth'):
    try:
        s=json.load(open(f'gpurun_out/r06/parity_sph_{v}.json'))['summary']
        print(v, {k:"%s %.2e [%.2e]"%(x['set'][:24],x['p99'],x['p99_hi']) for k,x in s['worst_det_not_exempt']['tile32'].items()}, 'median', {k:"%.2e"%x for k,x in s['median_of_p99_det'].items()}, 'fails', len(s['failures']), 'exempt', list(s['exempt_reference_fp32_also_above_bound']))
    except Exception as e: print(v,'ERR',e)
P
timeout 1500 python3 -m pytest tests/test_gpu_row_index.py tests/test_gpu_configs.py tests/test_gpu_wavefront.py -x -q > $O/t_wavefront.log 2>&1
echo "wavefront/config tests rc=$?"; tail -4 $O/t_wavefront.log
timeout 300 python3 tools/render_array.py --passes 64 > $O/render_array_direct.json 2>&1; tail -1 $O/render_array_direct.json
