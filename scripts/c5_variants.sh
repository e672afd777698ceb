#!/bin/bash
# GPU box: per-kernel times of phased config-5 passes (one 1000-step block: every phase with the chip to itself) for several
# builds of the library.  usage: scripts/c5_variants.sh <tag> lib1.so lib2.so ...   ("-" = the in-tree library)
set -u
TAG=${1:-c5var}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so); [ "$lib" = "-" ] && name=tree
  L=$lib; [ "$lib" != "-" ] && L=$R/$lib
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 $R/scripts/c5_pass.py $L 1000 1 > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; exit 1; }
  f=$(find $OUT/$name -name '*kernel_stats.csv' | head -1)
  echo "== $name" | tee -a $OUT/summary.txt
  python3 - "$f" <<'PY' | tee -a $OUT/summary.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'cont_' in r['Name']:
        print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
  rm -rf $OUT/$name
done
