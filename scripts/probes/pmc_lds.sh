#!/bin/bash
# LDS pipe pressure: scripts/pmc_lds.sh lib.so [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
lib=$1; shift
LIBARG="--lib $R/$lib"
OUT=$R/gpurun_out/pmc_lds/$(basename $lib .so)
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_lds -- python3 $R/bench.py $LIBARG --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic --no-extras "$@" > $OUT/log 2>&1
python3 $R/scripts/summarize_profile.py $OUT | grep -A9 "grid=1048576" | head -10
