#!/usr/bin/env python3
"""fp64 VALU issue rate this GPU sustains in practice (calibrates the compute roofline quoted in DESIGN.md):
rsik_debug_math op 6 runs 8 x 2048 v_fma_f64 per lane; prints wave-instructions/s and the equivalent clock."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reachy2_symbolic_ik_amd.backend import HipSolver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
hs = HipSolver(0)
a = torch.rand(n, dtype=torch.float64, device="cuda")
b = torch.full((n,), 0.999, dtype=torch.float64, device="cuda")
for _ in range(3):
    hs.debug_math(6, a, b)
torch.cuda.synchronize()
reps = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    hs.debug_math(6, a, b)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fma_wave_instr = (n / 64) * 8 * 2048
rate = fma_wave_instr / (ms * 1e-3)
print("n=%d  %.3f ms/launch  %.3e wave-FMA/s  = %.3f GHz-equivalent at 1 wave-instr / 4 cycles / SIMD (1024 SIMDs)"
      % (n, ms, rate, rate * 4 / 1024 / 1e9))
print("fp64 FMA TFLOP/s: %.1f" % (rate * 64 * 2 / 1e12))
