#!/bin/bash
# GPU box: per-wave timeline of solve_kernel (build/variants/probe_timeline.so = -DRSIK_TIMELINE_PROBE build),
# with and without device-resident kernarg buffers, then the bench line both ways.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/timeline; mkdir -p $OUT
PROBE=$R/build/variants/${1:-probe_timeline.so}
TAG=${2:-t}
timeout -k 10 120 python3 $R/scripts/timeline_probe.py --lib $PROBE > $OUT/${TAG}_1m.txt 2>&1 || { cat $OUT/${TAG}_1m.txt; exit 1; }
HIP_FORCE_DEV_KERNARG=1 timeout -k 10 120 python3 $R/scripts/timeline_probe.py --lib $PROBE > $OUT/${TAG}_1m_devkernarg.txt 2>&1
grep -v "^  " $OUT/${TAG}_1m.txt; grep -A9 "per XCD" $OUT/${TAG}_1m.txt
echo "=== HIP_FORCE_DEV_KERNARG=1"; grep -v "^  " $OUT/${TAG}_1m_devkernarg.txt
for i in 1 2 3; do
  for kv in 0 1; do
    HIP_FORCE_DEV_KERNARG=$kv timeout -k 10 120 python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('devkernarg=$kv', round(d['roofline']['kernel_ms']*1000,2),'us', round(d['value']/1e9,2),'G/s')"
  done
done
