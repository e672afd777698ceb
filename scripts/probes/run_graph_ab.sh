#!/bin/bash
# GPU box: eager pre-bound launches vs hipGraph replay of the K steps, at several K
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
for k in 50 200 1000; do
  for mode in "" "--graph"; do
    timeout -k 10 200 python3 $R/bench.py --steps $k --warmup 10 --no-cpu-baseline --no-extras $mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=$k mode=${mode:-eager}', 'kernel', round(d['roofline']['kernel_ms']*1000,2),'us  step', round(d['ms_per_step']*1000,2), 'us ', round(d['value']/1e9,2),'G/s')"
  done
done
done
