#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for lib in "$@"; do
  LIBARG="--lib $R/$lib"
  OUT=$R/gpurun_out/pmc_mix/$(basename $lib .so)
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $LIBARG --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --no-extras > $OUT/log 2>&1
  echo "== $lib"; python3 $R/scripts/summarize_profile.py $OUT | grep -A10 "grid=1048576" | head -10
done
