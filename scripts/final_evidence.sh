#!/bin/bash
# GPU box: everything profiles/<round>/ holds besides the profile.sh collections, for the tree as it is.
# usage: scripts/final_evidence.sh <tag>   -> gpurun_out/<tag>/...
set -u
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err || exit 1
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/drv_trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/drv_trace.log 2>&1) || exit 1
cp $(find $OUT/drv_trace -name '*kernel_stats.csv' | head -1) $OUT/driver_flags_kernel_stats.csv
rm -rf $OUT/drv_trace
python3 scripts/c5_ab.py > $OUT/c5_block_sweep.txt 2>&1 || exit 1
python3 scripts/c5_graph.py > $OUT/c5_graph.txt 2>&1 || exit 1
bash scripts/c5_trace.sh ${TAG}_tl - 0 > $OUT/c5_pipeline_timeline.txt 2>&1 || exit 1
bash scripts/c5_trace.sh ${TAG}_tl1 - 1000 > $OUT/c5_phases_alone.txt 2>&1 || exit 1
python3 scripts/scalar_latency.py > $OUT/scalar_latency.txt 2>&1 || exit 1
python3 scripts/soak_parity.py 1048576 2 > $OUT/soak_parity.txt 2>&1 || exit 1
[ -x build/issue_probe ] && timeout -k 5 120 ./build/issue_probe > $OUT/issue_probe.txt 2>&1
tail -3 $OUT/c5_block_sweep.txt; tail -3 $OUT/c5_graph.txt; tail -4 $OUT/scalar_latency.txt; tail -2 $OUT/soak_parity.txt
