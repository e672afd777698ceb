import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from reachy2_symbolic_ik_amd import ControlIK, _abi
n, n_steps = 4096, 1000
traj = bench.make_config5_trajectories(n, n_steps, seed=20250204, device=0)
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
cont = ctrl.new_continuous_state("r_arm", n)
out = {"joints": torch.empty((n_steps, n, 7), dtype=torch.float64, device="cuda"),
       "reachable": torch.empty((n_steps, n), dtype=torch.uint8, device="cuda"),
       "state": torch.empty((n_steps, n), dtype=torch.uint8, device="cuda")}
ctrl.run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=out)
torch.cuda.synchronize()
st = out["state"]
print("state histogram:", torch.bincount(st.flatten().long(), minlength=8).tolist())
print("cont rows emergency latched:", int((cont[9] != 0).sum()), "of", n)
j = out["joints"]
d = (j[1:] - j[:-1]).abs()
print("max |dj| per joint:", d.amax(dim=(0, 1)).tolist())
print("steps with |dj|>0.4:", int((d > 0.4).any(dim=2).sum()), "outside [-pi,pi]:", int((j.abs() > 3.141592653589793).any(dim=2).sum()), "nan:", int(torch.isnan(j).any(dim=2).sum()))
print("reachable frac:", float(out["reachable"].float().mean()))
