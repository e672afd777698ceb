"""Debug: G4 'var' cases through the scalar ControlIK path under the three grid-search strategies."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reachy2_symbolic_ik_amd import ControlIK, _abi
g = np.load("tests/golden/g4_control_discrete.npz")
c = ControlIK(urdf_path="config_files/reachy2_ik_minimal.urdf")
c.nb_search_points = 20
for arm in ("r_arm", "l_arm"):
    pre = f"std_{arm}_"
    M = g[pre + "M"]; idx = g[pre + "var_idx"]
    for k in range(0, len(idx), 4):
        row = []
        for mode in (1, 2, 0):
            c._solver.set_option(_abi.OPT_SWEEP_MODE, mode)
            j, ok, st = c.symbolic_inverse_kinematics(arm, M[idx[k]], "discrete", current_joints=list(g[pre + "var_current_joints"][k]),
                                                      preferred_theta=float(g[pre + "var_preferred_theta"][k]))
            err = float(np.max(np.abs(np.array(j) - g[pre + "var_joints"][k])))
            row.append((ok, st, err))
        bad = any(r[2] > 1e-9 or r[0] != bool(g[pre + "var_reachable"][k]) for r in row)
        if bad:
            print(arm, k, "pref", g[pre + "var_preferred_theta"][k], "gold reach", g[pre + "var_reachable"][k], row)
print("done")
