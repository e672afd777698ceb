#!/bin/bash
# A/B kernel variants in one GPU session: scripts/ab.sh libA.so libB.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
for lib in "$@"; do
  RSIK_LIB_PATH=$R/$lib timeout 120 python3 $R/bench.py --config ${CFG:-2} --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['roofline']['kernel_ms']*1000,2),'us', round(d['value']/1e9,2),'G/s')"
done; done
