#!/bin/bash
# Round 6 (docs/experiments.md R6.5): the pipeline's chip-filling kernels capped to 3 / 4 workgroups per compute unit by unused dynamic LDS
# so that a lone 241-register chain wave always finds room: long runs and K overlapping passes, library by library on one box, two rounds.
# The variant libraries were built from a two-line patch that is NOT in the tree (no gain: profiles/r06/config5_occupancy_caps_ab.txt): the
# third argument of the prepare / joints launches in rsik_control_continuous_run (dynamic LDS bytes, 0) replaced by macros, then
# scripts/build_variant.py occ_j3 -DRSIK_JOINTS_PAD_LDS=31232 (3 workgroups per compute unit; 18928: 4), -DRSIK_PREP_PAD_LDS=45568 (3; 33280: 4).
# usage: scripts/probes/c5_occupancy_ab.sh <out file> lib.so ...
out=$1; shift
mkdir -p "$(dirname "$out")"
: > "$out"
for round in 1 2; do
  for lib in "$@"; do
    echo "== round $round $lib" >> "$out"
    C5_LIB=$lib timeout -k 10 200 python scripts/probes/c5_long_runs.py 4096 --steps 4000,16000 --blocks 0 --reps 4 >> "$out" 2>&1 || exit 1
    C5_LIB=$lib timeout -k 10 200 python scripts/probes/c5_overlap.py 4096 1000 --rounds 2 --no-graph >> "$out" 2>&1 || exit 1
  done
done
