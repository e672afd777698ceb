#!/bin/bash
# config 5, eager and replayed, for several uniform block sizes (RSIK_OPT_CONT_BLOCK_STEPS): scripts/probes/c5_uneven_sweep.sh lib.so "0 256 336 512"
for b in ${2:-0 256 336 400 512}; do C5_BLOCK=$b timeout -k 10 120 python scripts/probes/c5_graph_libs.py $1 2>&1 | grep -v amdgpu.ids | sort | tail -1; done
