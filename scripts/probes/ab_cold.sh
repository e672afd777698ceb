#!/bin/bash
# A/B of library variants on warm (Infinity-Cache-resident) and cold (HBM) inputs, interleaved: scripts/ab_cold.sh libA.so libB.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=${ROUNDS:-3}
for round in $(seq $ROUNDS); do
for lib in "$@"; do
  timeout 200 python3 $R/bench.py --lib $R/$lib --config ${CFG:-2} --steps ${STEPS:-400} --warmup 20 --no-cpu-baseline --no-live-traffic --no-valu-calibration 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', 'warm', round(r['kernel_ms']*1000,2), 'cold', round(r['kernel_ms_cold']*1000,2), 'us')"
done; done
