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
# (config 5 under the counters: issued launch by launch with its streams tied by events, as in round 3 — a profiler that serialises
# dispatches deadlocks on a replayed multi-stream graph (round 4: every --pmc pass of `--launch graph` sat until its timeout), and
# waits on device words are not its business either)
LAUNCH="eager --no-steady-state"; [ "$CFG" = 5 ] && LAUNCH="eager --phased-variant 1 --no-steady-state"
python3 $R/bench.py --config $CFG > $OUT/bench.json 2> $OUT/bench.err
# (config 5 under the kernel trace: the captured form, as in round 5 — the default is now the pipelined one, whose streams are tied by
# device words and whose theta kernels wait for their prepare kernels themselves: not something to run under a tool that may serialise)
TRACE_LAUNCH=""; [ "$CFG" = 5 ] && TRACE_LAUNCH="--launch graph"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config $CFG $TRACE_LAUNCH --no-cpu-baseline --no-extras --no-steady-state --no-live-traffic > $OUT/trace.log 2>&1
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --config $CFG --steps 5 --warmup 2 --launch $LAUNCH --no-cpu-baseline --no-extras --no-live-traffic > $OUT/pmc_fetch.log 2>&1
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --config $CFG --steps 5 --warmup 2 --launch $LAUNCH --no-cpu-baseline --no-extras --no-live-traffic > $OUT/pmc_write.log 2>&1
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --config $CFG --steps 5 --warmup 2 --launch $LAUNCH --no-cpu-baseline --no-extras --no-live-traffic > $OUT/pmc_sq.log 2>&1
timeout -k 10 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_misc -- python3 $R/bench.py --config $CFG --steps 5 --warmup 2 --launch $LAUNCH --no-cpu-baseline --no-extras --no-live-traffic > $OUT/pmc_misc.log 2>&1
python3 $R/scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/bench.json; cat $OUT/summary.txt
