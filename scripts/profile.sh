#!/bin/bash
# Runs on the GPU box (via gpurun): bench lines + rocprofv3 kernel-trace stats + PMC passes for one config.
# usage: scripts/profile.sh <tag> <config>     -> gpurun_out/<tag>/...
set -u
TAG=${1:-prof}; CFG=${2:-2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python3 $R/bench.py --config $CFG > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config $CFG --no-cpu-baseline --no-extras > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --config $CFG --steps 5 --warmup 2 --launch eager --no-cpu-baseline --no-extras > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --config $CFG --steps 5 --warmup 2 --launch eager --no-cpu-baseline --no-extras > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --config $CFG --steps 5 --warmup 2 --launch eager --no-cpu-baseline --no-extras > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_misc -- python3 $R/bench.py --config $CFG --steps 5 --warmup 2 --launch eager --no-cpu-baseline --no-extras > $OUT/pmc_misc.log 2>&1
python3 $R/scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/bench.json; cat $OUT/summary.txt
