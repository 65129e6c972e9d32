#!/bin/bash
# Secondary measurements quoted in DESIGN.md / README.md, in one run on the GPU box:
#   tools/secondary_bench.sh <tag>   -> gpurun_out/secondary_<tag>.txt   (copy to profiles/)
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/secondary_$TAG.txt
cd $REPO
{
echo "# secondary measurements, $(date -u +%Y-%m-%dT%H:%MZ), $(python -c 'import torch;print(torch.cuda.get_device_name(0))' 2>/dev/null)"
echo "## config 4: 52 measured materials mixed in one 16 Mi wavefront (tools/bench_extra.py mixed)"
python tools/bench_extra.py mixed 2>/dev/null | tail -1
python tools/mixed_breakdown.py 2>/dev/null | grep ms
echo "## f2: reflow teacher sampler, 64x6 net, 4 Mi rows, T=128, no Jacobian, fp16 (tools/bench_extra.py teacher)"
python tools/bench_extra.py teacher 2>/dev/null | tail -1
echo "## config 5: 512^2 x 256 passes x 4 spp material ball (tools/bench_render.py)"
python tools/bench_render.py --passes 256 2>/dev/null | tail -1
python tools/bench_render.py --passes 256 --plugin spherical 2>/dev/null | tail -1
python tools/bench_render.py --passes 256 --material chm_orange_rgb --measured-dir tests/golden 2>/dev/null | tail -1
echo "## kernel time vs batch size, plugin pdf launches (tools/nscan.py)"
python tools/nscan.py 2>/dev/null | grep cl=
echo "## per-step / per-query split (tools/tscan.py)"
python tools/tscan.py 2>/dev/null | grep -v amdgpu
echo "## sample(wi) + pdf(wi, wl): two launches vs bsdfd_plugin_sample_pdf (tools/fused_ab.py)"
python tools/fused_ab.py 2>/dev/null | grep -v amdgpu
echo "## full plugin calls with the native ground truth, 1 Mi queries (tools/plugin_overhead.py)"
python tools/plugin_overhead.py 2>/dev/null | grep sample
echo "## stand-alone encoding pass (tools/enc_bench.py)"
python tools/enc_bench.py 2>/dev/null | grep N=
echo "## accuracy per precision mode on the golden cases, pdf error vs the fp64 oracle (tools/quick_gpu.py)"
python tools/quick_gpu.py 2>/dev/null | grep "x_err"
echo "## importance-sampling quality with the native ground truth (tools/is_quality.py)"
python tools/is_quality.py 2>/dev/null | grep wi=
echo "## 12-ball array scene, the reference's film size and sample count (tools/render_array.py)"
python tools/render_array.py --passes 256 --width 1366 --height 1024 --out gpurun_out/array0_full 2>/dev/null | tail -1
echo "## host call overhead (tools/host_overhead.py)"
python tools/host_overhead.py 2>/dev/null | grep N=
} > $OUT 2>&1
cat $OUT
