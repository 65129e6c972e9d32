#!/bin/bash
# full GPU suite + smoke.  gpurun -- bash tools/r04_tests.sh [extra pytest args]
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_tests
mkdir -p $OUT
cd $REPO
rm -f gpurun_out/plugin_parity.jsonl
python3 -m pytest tests -x -q -m gpu "$@" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -15 $OUT/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke rc=$?"; tail -3 $OUT/smoke.log
cp gpurun_out/plugin_parity.jsonl $OUT/ 2>/dev/null
