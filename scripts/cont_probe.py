#!/usr/bin/env python3
"""Where a control step's time goes in trajectory mode (config 5), from a -DRSIK_CONT_PROBE build (diagnostic only):

    hipcc ... -DRSIK_CONT_PROBE rsik_lib.hip -o build/variants/probe_cont.so
    python scripts/cont_probe.py --lib build/variants/probe_cont.so
"""
import os
import sys

import numpy as np
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
from reachy2_symbolic_ik_amd import ControlIK  # noqa: E402

n_traj, n_steps = 4096, 1000
traj = bench.make_config5_trajectories(n_traj, n_steps)
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF)
st = ctrl.new_continuous_state("r_arm", n_traj)
st0 = st.clone()
for _ in range(2):
    st.copy_(st0)
    ctrl.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0])
torch.cuda.synchronize()
pc = st[1:7, ::64].cpu().numpy() / n_steps          # core-clock ticks per step and phase, one column per wave
names = ["goal matrix / start-up", "reach (with limits)", "10-point grid + no-limits reach", "theta limit + joints",
         "safety checks + continuity", "stores"]
tot = pc.sum(axis=0)
print("waves %d, core-clock ticks per control step: median %.0f (p5 %.0f, p95 %.0f)" % (pc.shape[1], np.median(tot), np.percentile(tot, 5), np.percentile(tot, 95)))
for k, nm in enumerate(names):
    print("  %-34s %7.0f ticks  %5.1f %%" % (nm, np.median(pc[k]), 100 * np.median(pc[k]) / np.median(tot)))
