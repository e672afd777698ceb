#!/bin/bash
# GPU box: everything profiles/<round>/ holds besides the profile.sh collections, for the tree as it is.
# usage: scripts/final_evidence.sh <tag> [a|b|ab]   -> gpurun_out/<tag>/...   (two parts: a call on the GPU box is limited to 20 minutes)
set -u
TAG=${1:-final}
PART=${2:-ab}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
if [[ $PART == *a* ]]; then
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err || exit 1
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/drv_trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-other-configs --no-live-traffic > $OUT/drv_trace.log 2>&1) || exit 1
cp $(find $OUT/drv_trace -name '*kernel_stats.csv' | head -1) $OUT/driver_flags_all_launches_kernel_stats.csv
# the statistics of the 20 TIMED launches alone (the run also holds 5 warm-up launches and one untimed replay of the graph)
python3 - $(find $OUT/drv_trace -name '*kernel_trace.csv' | head -1) > $OUT/driver_flags_timed20_kernel_stats.txt <<'PY'
import csv, statistics, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "solve_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
t = d[-20:]
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows[-20:-1], rows[-19:])]
print(f"solve_kernel launches in the run: {len(d)} (5 warm-up, 20 of the untimed graph replay, 20 timed)")
print(f"the 20 timed launches: avg {statistics.mean(t):.2f} us  median {statistics.median(t):.2f}  min {min(t):.2f}  max {max(t):.2f}  stdev {statistics.pstdev(t):.2f}")
print(f"gaps between them: avg {statistics.mean(gaps):.2f} us  max {max(gaps):.2f}")
print(f"all launches:          avg {statistics.mean(d):.2f} us  median {statistics.median(d):.2f}  max {max(d):.2f}")
PY
rm -rf $OUT/drv_trace
C5_BLOCKS=0,64,128,192,256,352,512,1000,0 python3 scripts/c5_ab.py > $OUT/c5_block_sweep.txt 2>&1 || exit 1
python3 scripts/soak_pipeline.py 20000 203 > $OUT/soak_pipeline.txt 2>&1 || exit 1
python3 scripts/disc_timeline_probe.py --lib build/variants/probe_timeline.so 262144 > $OUT/discrete_timeline_262144.txt 2>&1
python3 scripts/c5_graph.py > $OUT/c5_graph.txt 2>&1 || exit 1
bash scripts/c5_trace.sh ${TAG}_tl - 0 > $OUT/c5_pipeline_timeline.txt 2>&1 || exit 1
bash scripts/c5_trace.sh ${TAG}_tl1 - 1000 > $OUT/c5_phases_alone.txt 2>&1 || exit 1
python3 scripts/scalar_latency.py > $OUT/scalar_latency.txt 2>&1 || exit 1
timeout -k 10 200 python3 scripts/probes/latched_pass.py 2>&1 | grep -v amdgpu > $OUT/c5_latched_passes.txt
python3 scripts/soak_parity.py 1048576 2 > $OUT/soak_parity.txt 2>&1 || exit 1
[ -x build/issue_probe ] && timeout -k 5 120 ./build/issue_probe > $OUT/issue_probe.txt 2>&1
tail -3 $OUT/c5_block_sweep.txt; tail -3 $OUT/c5_graph.txt; tail -4 $OUT/scalar_latency.txt; tail -2 $OUT/soak_parity.txt
fi
if [[ $PART == *b* ]]; then
# round 4: the stage timers, what an edge between streams costs, the other forms of the continuous run (bits, time, work-item trace),
# the pass without a profiler (in-kernel stamps), the bounds, eager against replayed at the driver's protocol
timeout -k 10 400 python3 scripts/stage_timers.py --no-build > $OUT/stage_timers.json 2> $OUT/stage_timers.err
[ -x build/edge_probe ] && timeout -k 5 120 ./build/edge_probe > $OUT/edge_probe.txt 2>&1
[ -f build/variants/pipe_timing.so ] && timeout -k 10 300 python3 scripts/probes/c5_untraced_timeline.py build/variants/pipe_timing.so > $OUT/c5_untraced_timeline.txt 2>&1
[ -f build/variants/disc_class.so ] && timeout -k 10 300 python3 scripts/probes/disc_sorted_bound.py --no-build > $OUT/disc_sorted_bound.txt 2>&1
timeout -k 10 400 bash scripts/probes/c5_forms.sh > $OUT/c5_eager_vs_replayed_driver_protocol.txt 2>&1
timeout -k 10 300 bash scripts/probes/c3_forms.sh > $OUT/c3_eager_vs_replayed_driver_protocol.txt 2>&1
tail -4 $OUT/edge_probe.txt; tail -3 $OUT/c5_eager_vs_replayed_driver_protocol.txt
fi
