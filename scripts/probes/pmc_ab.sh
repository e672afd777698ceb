#!/bin/bash
# PMC comparison of kernel variants: scripts/pmc_ab.sh libA.so libB.so ...
# (config 2 only, every pass under a time limit: the default protocol would also issue config 5, whose device-word waits a
# serialising profiler deadlocks on — bench.live_traffic and scripts/profile.sh force --phased-variant 1 for it)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for lib in "$@"; do
  LIBARG="--lib $R/$lib"
  OUT=$R/gpurun_out/pmc_ab/$(basename $lib .so)
  rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $LIBARG --config 2 --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic --no-extras --no-steady-state > $OUT/log 2>&1
  timeout -k 10 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 --kernel-trace --output-format csv -d $OUT/pmc_misc -- python3 $R/bench.py $LIBARG --config 2 --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic --no-extras --no-steady-state >> $OUT/log 2>&1
  python3 $R/scripts/summarize_profile.py $OUT | grep -A30 "grid=1048576" | grep -v "^pmc.*4194304" | head -24
done
