#!/bin/bash
# config-5 passes of the single launch under several cuts of the work (block / prepare steps / joints sub-groups / look-ahead / chain lag)
for cfg in "${@:-64 8 8 8 4}"; do
  set -- $cfg
  echo "== S=$1 SP=$2 JH=$3 L=$4 CL=$5"
  FUSED_S=$1 FUSED_SP=$2 FUSED_JH=$3 FUSED_L=$4 FUSED_CL=$5 timeout -k 10 200 python scripts/fused_check.py 4096 1000 10 2>&1 | grep -E 'fused   eager|prepare |joints |chain  |theta block|MISMATCH|block  (0|15)'
done
