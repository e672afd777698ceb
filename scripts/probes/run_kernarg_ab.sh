#!/bin/bash
# GPU box: does the default set by bench.py / the package (HIP_FORCE_DEV_KERNARG=1 at import) reach the HIP runtime?
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do
  for kv in unset 0 1; do
    if [ $kv = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$kv; fi
    timeout -k 10 120 python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('HIP_FORCE_DEV_KERNARG=$kv', round(d['roofline']['kernel_ms']*1000,2),'us', round(d['value']/1e9,2),'G/s')"
  done
done
