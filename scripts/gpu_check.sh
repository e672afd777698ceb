#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, the default bench line and smoke().  usage: scripts/gpu_check.sh <tag>
set -u
TAG=${1:-check}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
rc=$?
echo "pytest exit $rc" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit $?
cat $OUT/bench_default.json
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee $OUT/smoke.log
