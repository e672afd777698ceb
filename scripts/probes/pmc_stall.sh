#!/bin/bash
# Where do the non-VALU cycles go: instruction fetch, I-cache, wave launch.  scripts/pmc_stall.sh lib.so [bench args]
# (config 2 only, every pass under a time limit: the default protocol would also issue config 5, whose device-word waits a
# serialising profiler deadlocks on — bench.live_traffic and scripts/profile.sh force --phased-variant 1 for it)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
lib=$1; shift
LIBARG="--lib $R/$lib"
OUT=$R/gpurun_out/pmc_stall/$(basename $lib .so)
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 150 rocprofv3 --pmc SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $LIBARG --config 2 --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic --no-extras --no-steady-state "$@" > $OUT/log 2>&1
timeout -k 10 150 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_TC_STALL --kernel-trace --output-format csv -d $OUT/pmc_sqc -- python3 $R/bench.py $LIBARG --config 2 --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic --no-extras --no-steady-state "$@" >> $OUT/log 2>&1
timeout -k 10 150 rocprofv3 --pmc SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_REQ_NO_ALLOC_CSN SPI_CSN_BUSY SPI_CSN_WAVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_spi -- python3 $R/bench.py $LIBARG --config 2 --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic --no-extras --no-steady-state "$@" >> $OUT/log 2>&1
python3 $R/scripts/summarize_profile.py $OUT | grep -A40 "grid=1048576" | grep -B40 -m2 "^pmc" | head -40
