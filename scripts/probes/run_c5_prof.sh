#!/bin/bash
# GPU box: config 5 (continuous mode) kernel time vs launch gaps
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/c5; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
python3 $R/bench.py --config 5 --steps 5 --warmup 2 > $OUT/bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
d = json.load(open(out + "/bench.json"))
print("bench: %.1f M steps/s, %.2f us per launch" % (d["value"] / 1e6, d["roofline"]["kernel_ms"] * 1e3))
for f in glob.glob(out + "/trace/**/*_kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "continuous" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
    gaps = sorted(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:]))
    print("continuous kernel launches %d: duration median %.2f us (p5 %.2f, p95 %.2f); gap to next launch median %.2f us (p95 %.2f)" % (
        len(dur), dur[len(dur) // 2] / 1e3, dur[len(dur) // 20] / 1e3, dur[-len(dur) // 20] / 1e3, gaps[len(gaps) // 2] / 1e3, gaps[-len(gaps) // 20] / 1e3))
    print("vgpr/lds/scratch:", rows[0]["VGPR_Count"], rows[0]["LDS_Block_Size"], rows[0]["Scratch_Size"])
PY
