#!/bin/bash
# GPU box: config-2 kernel time vs batch size (bench.py --poses N, default launch mode) -> gpurun_out/size_sweep.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/size_sweep.txt
echo "# config 2, poses per launch : us per launch : G solves/s : fraction of 8 TB/s at 122 B/pose" > $OUT
for n in 65536 262144 524288 1048576 2097152 4194304 8388608 16777216; do
  k=$(( n <= 1048576 ? 1000 : (n <= 4194304 ? 300 : 100) ))
  timeout -k 10 200 python3 $R/bench.py --poses $n --steps $k --warmup 10 --no-cpu-baseline --no-live-traffic --no-extras 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%9d : %8.2f : %6.2f : %.3f' % ($n, r['kernel_ms'] * 1e3, d['value'] / 1e9, r['frac']))" | tee -a $OUT
done
