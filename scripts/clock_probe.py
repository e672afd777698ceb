#!/usr/bin/env python3
"""Core clock and wave lifetime of solve_kernel under load, from a -DRSIK_CLOCK_PROBE build (diagnostic only):

    hipcc ... -DRSIK_CLOCK_PROBE rsik_lib.hip -o probe.so;  python scripts/clock_probe.py --lib probe.so
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def _lib_arg():
    """--lib PATH: the probe build to load instead of the in-tree library (must happen before the package loads it)."""
    if "--lib" in sys.argv:
        k = sys.argv.index("--lib")
        from reachy2_symbolic_ik_amd import _abi

        _abi.use_library(sys.argv[k + 1])
        del sys.argv[k: k + 2]


_lib_arg()
import bench  # noqa: E402
from reachy2_symbolic_ik_amd import SymbolicIK  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
ik = SymbolicIK("r_arm")
import numpy as np  # noqa: E402

P, E = bench.make_config2_poses(n)
poses = torch.as_tensor(np.ascontiguousarray(np.concatenate([P.T, E.T], axis=0))).cuda()
for _ in range(5):
    res = ik.solver.solve(poses, arm_uniform=0)
torch.cuda.synchronize()
iv = res["interval"][::64].cpu().numpy()
core, real = iv[:, 0], iv[:, 1]
print("waves", len(core), "core ticks/wave median %.0f" % float(sorted(core)[len(core) // 2]),
      "100MHz ticks/wave median %.0f" % float(sorted(real)[len(real) // 2]))
print("core clock under load: %.3f GHz" % (core.sum() / real.sum() * 0.1))
print("wave lifetime: median %.2f us, p5 %.2f, p95 %.2f" % tuple(float(x) / 100 for x in
      (sorted(real)[len(real) // 2], sorted(real)[len(real) // 20], sorted(real)[-len(real) // 20])))
