#!/usr/bin/env python3
"""Where the phased pipeline and the step kernel part ways: first differing (step, trajectory) per state code, with the numbers.
usage: handover_debug.py [trajectories] [steps]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from reachy2_symbolic_ik_amd import _abi as A
if os.environ.get("LIB"):
    A.use_library(os.path.abspath(os.environ["LIB"]))
from reachy2_symbolic_ik_amd import ControlIK

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
traj = bench.make_config5_trajectories(n, T, seed=20250204, device=0)
c = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
c._solver.set_option(A.OPT_EULER_ROUNDTRIP, int(os.environ.get("EULER", "0")))
res = {}
for name, mode in (("steps", A.CONT_RUN_STEPS), ("phased", A.CONT_RUN_PHASED)):
    c._solver.set_option(A.OPT_CONT_RUN_MODE, mode)
    st = c.new_continuous_state("r_arm", n)
    o = c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0])
    torch.cuda.synchronize()
    res[name] = {k: v.cpu().numpy() for k, v in o.items()}
    res[name]["cont"] = st.cpu().numpy()
a, b = res["steps"], res["phased"]
print("state equal:", np.array_equal(a["state"], b["state"]), " reachable equal:", np.array_equal(a["reachable"], b["reachable"]))
dj = np.abs(a["joints"] - b["joints"])
dj = np.nan_to_num(dj, nan=0.0).max(axis=2)   # [T, n]
print("max joint diff", dj.max(), " fraction of steps > 1e-9:", (dj > 1e-9).mean())
first = np.full(n, T)
for i in range(n):
    w = np.flatnonzero(dj[:, i] > 1e-9)
    if w.size:
        first[i] = w[0]
bad = np.flatnonzero(first < T)
print("trajectories that differ:", bad.size, "of", n, " latched at the end (steps / phased):", int((a["cont"][9] != 0).sum()), int((b["cont"][9] != 0).sum()))
for i in bad[:12]:
    t = first[i]
    print(f"  trajectory {i}: first differs at step {t} (state {a['state'][t, i]}), |d joints| {np.abs(a['joints'][t, i] - b['joints'][t, i]).round(4)}; "
          f"step before: states {a['state'][t - 1, i]} / {b['state'][t - 1, i]}; m20 {float(traj[t, 6, i]):.12f}")
codes = {}
for i in bad[:2000]:
    t = first[i]
    codes.setdefault(int(a["state"][t, i]), []).append((t, i))
for code, lst in sorted(codes.items()):
    t, i = lst[0]
    print(f"state code {code}: {len(lst)} first differences, e.g. step {t} trajectory {i}")
    print("   steps :", np.array2string(a["joints"][t, i], precision=6), "reach", a["reachable"][t, i])
    print("   phased:", np.array2string(b["joints"][t, i], precision=6), "reach", b["reachable"][t, i], "state", b["state"][t, i])
    print("   goal pos:", traj[t, 9:12, i].cpu().numpy())
