#!/usr/bin/env python3
"""What every entry point does today with NaN / Inf in a row of its input: the poisoned row's outputs, whether its neighbours' bits
change against a clean run, whether anything stalls (the run is bounded by the caller's `timeout`)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from reachy2_symbolic_ik_amd import ControlIK, SymbolicIK, _abi as A  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)
n = 4096
pos = bench.SHOULDER_R + rng.uniform(-0.5, 0.5, size=(n, 3))
eul = rng.uniform(-np.pi, np.pi, size=(n, 3))
soa = torch.as_tensor(np.ascontiguousarray(np.concatenate([pos.T, eul.T], axis=0))).to(dev)
ik = bench._quiet(SymbolicIK, "r_arm", device=0)
bad_rows = [5, 70, 700, 2049, 4095]
poisons = [("nan x", 0, np.nan), ("inf y", 1, np.inf), ("-inf z", 2, -np.inf), ("nan roll", 3, np.nan), ("inf yaw", 5, np.inf)]


def same_except(a, b, rows):
    keep = torch.ones(a.shape[0], dtype=torch.bool, device=a.device)
    keep[rows] = False
    a2, b2 = a[keep], b[keep]
    if a2.dtype == torch.float64:
        return torch.equal(a2.view(torch.int64), b2.view(torch.int64))
    return torch.equal(a2, b2)


clean = {k: v.clone() for k, v in ik.solve_batch(soa).items()}
p = soa.clone()
for (nm, comp, val), r in zip(poisons, bad_rows):
    p[comp, r] = val
out = ik.solve_batch(p)
torch.cuda.synchronize()
print("rsik_solve: neighbours identical:", all(same_except(out[k], clean[k], bad_rows) for k in clean))
for (nm, comp, val), r in zip(poisons, bad_rows):
    print(f"  {nm:9s} row {r}: reachable {int(out['reachable'][r])} state {int(out['state'][r])} (clean {int(clean['state'][r])}) joints {out['joints'][r].cpu().numpy()[:3]} interval {out['interval'][r].cpu().numpy()}")

# discrete
M = bench.make_config3_matrices(n, device=0)
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
clean = {k: v.clone() for k, v in ctrl.symbolic_inverse_kinematics_batch("r_arm", M).items()}
Mp = M.copy()
mpo = [("nan R00", (0, 0), np.nan), ("inf tx", (0, 3), np.inf), ("nan tz", (2, 3), np.nan), ("inf R12", (1, 2), np.inf), ("-inf ty", (1, 3), -np.inf)]
for (nm, ij, val), r in zip(mpo, bad_rows):
    Mp[r][ij] = val
out = ctrl.symbolic_inverse_kinematics_batch("r_arm", Mp)
torch.cuda.synchronize()
print("rsik_control_discrete: neighbours identical:", all(same_except(out[k], clean[k], bad_rows) for k in clean))
for (nm, ij, val), r in zip(mpo, bad_rows):
    print(f"  {nm:9s} row {r}: reachable {int(out['reachable'][r])} state {int(out['state'][r])} (clean {int(clean['state'][r])}) emergency {int(out['emergency'][r])} joints {out['joints'][r].cpu().numpy()[:3]}")
cj = np.zeros((n, 7))
cj[bad_rows[0], 2] = np.nan
cj[bad_rows[1], 0] = np.inf
out2 = ctrl.symbolic_inverse_kinematics_batch("r_arm", M, current_joints=cj)
clean2 = ctrl.symbolic_inverse_kinematics_batch("r_arm", M, current_joints=np.zeros((n, 7)))
torch.cuda.synchronize()
print("rsik_control_discrete current_joints: neighbours identical:", all(same_except(out2[k], clean2[k], bad_rows[:2]) for k in clean2))
for r in bad_rows[:2]:
    print(f"  row {r}: reachable {int(out2['reachable'][r])} state {int(out2['state'][r])} (clean {int(clean2['state'][r])}) joints {out2['joints'][r].cpu().numpy()[:3]}")

# continuous run, every form
n_traj, n_steps = 300, 200
traj = bench.make_config5_trajectories(n_traj, n_steps, seed=9, device=0)
hs = ctrl._solver
cont0 = ctrl.new_continuous_state("r_arm", n_traj)
bad_t = [3, 64, 130, 299]
tp = traj.clone()
tp[50, 9, bad_t[0]] = float("nan")       # tx at step 50
tp[0, 0, bad_t[1]] = float("inf")        # R00 at step 0
tp[120:, 11, bad_t[2]] = float("nan")    # tz from step 120 on
tp[199, 4, bad_t[3]] = float("-inf")     # last step
for name, mode in (("steps", A.CONT_RUN_STEPS), ("phased", A.CONT_RUN_PHASED)):
    hs.set_option(A.OPT_CONT_RUN_MODE, mode)
    res = {}
    for tag, T in (("clean", traj), ("bad", tp)):
        st = cont0.clone()
        o = ctrl.run_continuous_trajectories("r_arm", T, st, first_step_timed_out=True, current_pose=T[0])
        hs.synchronize()
        res[tag] = ({k: v.clone() for k, v in o.items()}, st.clone())
    keep = torch.ones(n_traj, dtype=torch.bool, device=dev)
    keep[bad_t] = False
    same = all(torch.equal(res["clean"][0][k][:, keep].contiguous().view(torch.uint8), res["bad"][0][k][:, keep].contiguous().view(torch.uint8)) for k in res["clean"][0])
    same = same and torch.equal(res["clean"][1][:, keep].contiguous().view(torch.uint8), res["bad"][1][:, keep].contiguous().view(torch.uint8))
    print(f"continuous_run[{name}]: neighbours identical: {same}")
    o, st = res["bad"]
    for t, steps in zip(bad_t, ([49, 50, 51, 199], [0, 1, 199], [119, 120, 199], [198, 199])):
        print(f"  traj {t}: " + "; ".join(f"step {s}: r {int(o['reachable'][s, t])} st {int(o['state'][s, t])} j0 {float(o['joints'][s, t, 0]):.4f}" for s in steps)
              + f" | prev_theta {float(st[0, t]):.4f}")
hs.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_AUTO)
# poisoned cont_state
st = cont0.clone()
st[0, 7] = float("nan")
o = ctrl.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=False)
hs.synchronize()
print("cont_state previous_theta NaN, traj 7: state codes", o["state"][:5, 7].cpu().numpy(), "joints finite:", bool(torch.isfinite(o["joints"][:, 7]).all()))
print("done")
