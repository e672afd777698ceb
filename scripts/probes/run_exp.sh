#!/bin/bash
# GPU box: A/B of build/variants/*.so given as arguments, then the timeline of PROBE (env) if set
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/exp; mkdir -p $OUT
export HIP_FORCE_DEV_KERNARG=1
ROUNDS=${ROUNDS:-3} $R/scripts/ab.sh "$@" 2>&1 | tee $OUT/ab.txt | grep "=="
if [ -n "${PROBE:-}" ]; then
  timeout -k 10 120 python3 $R/scripts/timeline_probe.py --lib $R/build/variants/$PROBE > $OUT/timeline_$PROBE.txt 2>&1
  grep -v "^  " $OUT/timeline_$PROBE.txt
fi
