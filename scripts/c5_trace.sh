#!/bin/bash
# GPU box: kernel trace of config-5 passes -> timeline.  usage: scripts/c5_trace.sh <tag> [lib|-] [steps per block] [run mode]
set -u
TAG=${1:-c5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
LIB=${2:--}
case "$LIB" in -|/*) ;; *) LIB=$R/$LIB ;; esac
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/scripts/c5_pass.py $LIB ${3:-0} ${4:-1} > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 $R/scripts/c5_timeline.py $f | tee $OUT/timeline.txt
rm -rf $OUT/trace
