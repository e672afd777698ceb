#!/bin/bash
# config 3 at the driver's protocol (W = 5, K = 20), issued eagerly against replayed from a hipGraph, alternating processes
for i in 1 2 3; do for f in eager graph; do python bench.py --config 3 --steps 20 --warmup 5 --launch $f --no-cpu-baseline --no-live-traffic --no-extras --no-other-configs 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$f', 'ms_per_step %.5f'%l['ms_per_step'], 'kernel_ms %.5f'%l['roofline']['kernel_ms'], 'frac %.3f'%l['roofline']['frac'])"; done; done
