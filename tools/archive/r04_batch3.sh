#!/bin/bash
# Round 4, GPU batch 3: wavefront pipeline (parity test + mixed_16Mi serial vs pipelined), counter reading of a pure-VALU stream
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r04; mkdir -p $O
python3 -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "pipeline" > $O/pipe_test.log 2>&1; echo "pipeline test rc=$?"; tail -3 $O/pipe_test.log
for r in 1 2; do
  BSDFD_BENCH_MIXED_SERIAL=1 python3 bench.py --workload mixed_16Mi --no-cpu-baseline --no-secondary > $O/mixed_serial_$r.json 2> $O/mixed_serial_$r.err
  python3 bench.py --workload mixed_16Mi --no-cpu-baseline --no-secondary > $O/mixed_pipe_$r.json 2> $O/mixed_pipe_$r.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/mixed_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(f, "value %.1f ms_per_step %.3f passes/step %d kernel ms/pass %.3f frac %.4f" % (d["value"], d["ms_per_step"], d["config"]["passes_per_step"], r["avg_launch_ms"]*4, r["frac"]))
    except Exception as e:
        print(f, "ERR", e)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_valu -o pmc -- $O/../../tools/ubench/mfma_src pmc0 > $O/pmc_valu.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/pmc_valu_grbm -o pmc -- $O/../../tools/ubench/mfma_src pmc0 > $O/pmc_valu_grbm.log 2>&1
tail -2 $O/pmc_valu.log
