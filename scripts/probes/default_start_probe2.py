#!/usr/bin/env python3
"""Diagnostic: G6 (default start) through the continuous-mode kernel: carried theta per step against the golden."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from reachy2_symbolic_ik_amd import ControlIK  # noqa: E402

np.set_printoptions(precision=17, linewidth=200)
g = np.load(os.path.join(ROOT, "tests", "golden", "g6_control_continuous.npz"))
for arm in ("r_arm", "l_arm"):
    c = bench._quiet(ControlIK, urdf_path=bench.URDF)
    print(arm, "constructor previous_theta", repr(c.previous_theta[arm]))
    Ms, TH, J = g[f"{arm}_M"], g[f"{arm}_previous_theta"], g[f"{arm}_joints"]
    nt = Ms.shape[0]
    st = c.new_continuous_state(arm, nt)
    prev = np.tile(np.asarray(c.previous_pose[arm], dtype=np.float64), (nt, 1, 1))
    for i in range(4):
        res = c.symbolic_inverse_kinematics_continuous_batch(arm, Ms[:, i], st, timed_out=np.full(nt, 1 if i == 0 else 0, dtype=np.uint8),
                                                             current_pose=prev)
        prev = Ms[:, i]
        th = st[0].cpu().numpy()
        print(" step", i, "theta gpu", th[:3], "golden", TH[:3, i], "state", res["state"].cpu().numpy()[:3])
        print("    max |dtheta|", np.max(np.abs(th - TH[:, i])), "max |djoints|", np.max(np.abs(res["joints"].cpu().numpy() - J[:, i])))
