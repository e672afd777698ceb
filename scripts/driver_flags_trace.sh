#!/bin/bash
# GPU box: `rocprofv3 --kernel-trace --stats` of the DRIVER'S command (python bench.py --steps 20 --warmup 5: the headline leg alone, no
# CPU legs) and the statistics of its 20 timed launches.  usage: scripts/driver_flags_trace.sh <tag>   -> gpurun_out/<tag>/
set -u
TAG=${1:-drv}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
(cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/drv_trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-other-configs --no-live-traffic --no-steady-state > $OUT/drv_trace.log 2>&1) || { tail -5 $OUT/drv_trace.log; exit 1; }
grep '^{' $OUT/drv_trace.log | tail -1 > $OUT/bench_driver_flags_under_rocprofv3.json
cp $(find $OUT/drv_trace -name '*kernel_stats.csv' | head -1) $OUT/driver_flags_all_launches_kernel_stats.csv
python3 - $(find $OUT/drv_trace -name '*kernel_trace.csv' | head -1) > $OUT/driver_flags_timed20_kernel_stats.txt <<'PY'
import csv, statistics, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "solve_kernel<0" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
big = [r for r in rows if int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0) >= 1048576] or rows
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in big]
t = d[-20:]
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(big[-20:-1], big[-19:])]
print(f"1 Mi-pose solve_kernel launches in the run: {len(d)} (workload filter aside: 5 warm-up, 20 of the untimed graph replay, 20 timed)")
print(f"the 20 timed launches: avg {statistics.mean(t):.2f} us  median {statistics.median(t):.2f}  min {min(t):.2f}  max {max(t):.2f}  stdev {statistics.pstdev(t):.2f}")
print(f"gaps between them: avg {statistics.mean(gaps):.2f} us  max {max(gaps):.2f}")
print(f"all of them:           avg {statistics.mean(d):.2f} us  median {statistics.median(d):.2f}  max {max(d):.2f}")
PY
rm -rf $OUT/drv_trace
cat $OUT/driver_flags_timed20_kernel_stats.txt
python3 -c "
import json,sys
d=json.loads(open('$OUT/bench_driver_flags_under_rocprofv3.json').read())
print('the line of the traced run: ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])
"
