#!/bin/bash
# A/B of two source trees in one GPU session, interleaved (for changes that alter the ABI, where scripts/ab.sh's --lib swap
# does not apply): scripts/ab_trees.sh <treeA> <treeB> ...   e.g. build/ab_base .     (CFG=3 for config 3)
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=${ROUNDS:-5}
LOG=$(mktemp)
for round in $(seq $ROUNDS); do
for tree in "$@"; do
  (cd $R/$tree && timeout 120 python3 bench.py --config ${CFG:-2} --steps ${STEPS:-400} --warmup 20 --no-cpu-baseline --no-live-traffic --no-extras 2>/dev/null) | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tree', round(d['roofline']['kernel_ms']*1000,2),'us', round(d['value']/1e9,2),'G/s')" | tee -a $LOG
done; done
python3 - $LOG <<'PY'
import sys, collections, statistics
t = collections.defaultdict(list)
for line in open(sys.argv[1]):
    p = line.split()
    if len(p) >= 2:
        t[p[0]].append(float(p[1]))
for k, v in t.items():
    print("== %-28s min %.2f  median %.2f us  (n=%d)" % (k, min(v), statistics.median(v), len(v)))
PY
rm -f $LOG
