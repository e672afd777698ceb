#!/bin/bash
# GPU box: what profiles/r06/ holds besides the profile.sh collections (round 6).  usage: scripts/r06_evidence.sh <tag> [a|b|c]
set -u
TAG=${1:-r06_ev}
PART=${2:-abc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
if [[ $PART == *a* ]]; then
# the driver's command, the config-5 probes of the round
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default_driver_flags.json 2> $OUT/bench_default_driver_flags.err || exit 1
timeout -k 10 300 python3 scripts/probes/c5_long_runs.py 4096 --steps 1000,2000,4000,8000,16000 --blocks 0,256,352,512 --variants 0 > $OUT/config5_long_runs.txt 2>&1
timeout -k 10 300 python3 scripts/probes/c5_overlap.py 4096 1000 --k 20 --rounds 3 > $OUT/config5_overlapping_passes.txt 2>&1
for v in 4 8 16; do echo "== RSIK_OPT_CONT_PHASED_VARIANT $v (4: the next run's prepare kernels not held at all; 8: held until the previous run's last chain kernel has FINISHED; 16: theta kernels behind stream waits)"; timeout -k 10 200 python3 scripts/probes/c5_overlap.py 4096 1000 --k 20 --rounds 2 --variant $v --no-graph; done > $OUT/config5_overlapping_passes_variants.txt 2>&1
timeout -k 10 200 python3 scripts/probes/c5_pass_sequence.py 300 1 > $OUT/config5_pass_sequence.txt 2>&1
timeout -k 10 200 python3 scripts/probes/c5_pass_sequence.py 300 0 >> $OUT/config5_pass_sequence.txt 2>&1
[ -f build/variants/pipe_timing.so ] && C5_GRAPH=0 C5_RESIDENT=1 RSIK_PIPE_TIMING_PRINT=1 timeout -k 10 200 python3 scripts/probes/c5_untraced_timeline.py build/variants/pipe_timing.so 0 1000 > $OUT/config5_overlapping_passes_timeline.txt 2>&1
[ -f build/variants/pipe_timing.so ] && C5_RESIDENT=0 RSIK_PIPE_TIMING_PRINT=1 timeout -k 10 200 python3 scripts/probes/c5_untraced_timeline.py build/variants/pipe_timing.so 0 1000 > $OUT/config5_untraced_timeline.txt 2>&1
[ -f build/variants/pipe_timing.so ] && C5_GRAPH=0 RSIK_PIPE_TIMING_PRINT=1 timeout -k 10 200 python3 scripts/probes/c5_untraced_timeline.py build/variants/pipe_timing.so 512 16384 > $OUT/config5_long_run_timeline.txt 2>&1
timeout -k 10 240 python3 scripts/probes/cu_mask_probe.py > $OUT/cu_mask_probe.txt 2>&1
tail -3 $OUT/config5_overlapping_passes.txt
fi
if [[ $PART == *b* ]]; then
# soaks, latched passes, scalar latencies
timeout -k 10 500 python3 scripts/soak_pipeline.py 20000 203 > $OUT/soak_pipeline.txt 2>&1 || { tail -5 $OUT/soak_pipeline.txt; exit 1; }
timeout -k 10 400 python3 scripts/soak_parity.py 1048576 2 > $OUT/soak_parity.txt 2>&1 || exit 1
timeout -k 10 200 python3 scripts/probes/latched_pass.py 2>&1 | grep -v amdgpu > $OUT/config5_latched_passes.txt
timeout -k 10 200 python3 scripts/scalar_latency.py > $OUT/scalar_latency.txt 2>&1
tail -2 $OUT/soak_pipeline.txt; tail -2 $OUT/soak_parity.txt
fi
if [[ $PART == *c* ]]; then
# the SCALE command's shape at the largest rank count a one-GPU box of this pool admits (its process guard: six processes on the card):
# config 4 at full size per rank, both gather forms
timeout -k 10 500 python3 bench.py --gpus 6 --backend gloo --single-device --steps 3 --warmup 1 --cpu-seconds 3 > $OUT/bench_six_ranks_gloo_gather_final.json 2> $OUT/bench_six_ranks_gloo_gather_final.err
# (the every-step form with eight stripes in a run of its own did not get past its set-up within seven minutes on the box that ran the
# line above in two — not pursued: the line above times BOTH forms, the second with four stripes per shard)
tail -c 600 $OUT/bench_six_ranks_gloo_gather_final.json; tail -3 $OUT/bench_six_ranks_gloo_gather_final.err
fi
