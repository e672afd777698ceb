#!/bin/bash
# GPU box: hardware counters of config-5 passes, the single launch (mode 3) beside the phased pipeline (mode 1, one block).
# usage: scripts/fused_pmc.sh <tag> [lib|-]
set -u
TAG=${1:-fpmc}; LIB=${2:--}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
L=$LIB; [ "$LIB" != "-" ] && L=$R/$LIB
run() {  # name mode blk counters...
  name=$1; mode=$2; blk=$3; shift 3
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $R/scripts/c5_pass.py $L $blk $mode > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; return 1; }
  f=$(find $OUT/$name -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$name" <<'PY' | tee -a $OUT/summary.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if 'cont_' not in k: continue
    acc[(k[:48], r['Counter_Name'])].append(float(r['Counter_Value']))
print('==', sys.argv[2])
for (k, c), v in sorted(acc.items()):
    print(f"  {k:48s} {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
  rm -rf $OUT/$name
}
for mode in 3 1; do
  blk=0; [ $mode = 1 ] && blk=1000
  run sq_m$mode $mode $blk SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
  run mem_m$mode $mode $blk SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT
  run ic_m$mode $mode $blk SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL GRBM_GUI_ACTIVE
done
