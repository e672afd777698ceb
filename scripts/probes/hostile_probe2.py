#!/usr/bin/env python3
"""Which neighbours of a poisoned trajectory change in the phased pipeline, in which arrays and at which steps."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from reachy2_symbolic_ik_amd import ControlIK, _abi as A  # noqa: E402

dev = torch.device("cuda", 0)
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
hs = ctrl._solver
n_traj, n_steps = 300, 200
traj = bench.make_config5_trajectories(n_traj, n_steps, seed=9, device=0)
cont0 = ctrl.new_continuous_state("r_arm", n_traj)
cases = {"nan tx step 50 traj 3": (50, 9, 3), "inf R00 step 0 traj 64": (0, 0, 64), "-inf R11 step 199 traj 299": (199, 4, 299)}
hs.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_PHASED)


def run(T):
    st = cont0.clone()
    o = ctrl.run_continuous_trajectories("r_arm", T, st, first_step_timed_out=True, current_pose=T[0])
    hs.synchronize()
    return {k: v.clone() for k, v in o.items()}, st.clone()


clean, cst = run(traj)
for name, (s, c, t) in cases.items():
    tp = traj.clone()
    tp[s, c, t] = float("nan") if "nan" in name else float("inf") if name.startswith("inf") else float("-inf")
    o, st = run(tp)
    print(name)
    for k in o:
        a, b = o[k], clean[k]
        d = (a.view(torch.int64) != b.view(torch.int64)) if a.dtype == torch.float64 else (a != b)
        if d.dim() == 3:
            d = d.any(dim=2)
        d[:, t] = False
        idx = d.nonzero()
        print(f"  {k}: {len(idx)} neighbour cells differ", idx[:6].cpu().numpy().tolist())
        if len(idx) and a.dtype == torch.float64:
            ss, tt = (int(v) for v in idx[0])
            print("    e.g.", a[ss, tt].cpu().numpy(), "vs", b[ss, tt].cpu().numpy())
    d = (st.view(torch.int64) != cst.view(torch.int64))
    d[:, t] = False
    print("  cont_state neighbour cells differ:", d.nonzero()[:6].cpu().numpy().tolist())
