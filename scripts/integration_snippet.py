"""The batched entry points exactly as INTEGRATION.md section 1 lists them, run once (tests/test_gpu_parity.py runs this file)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from reachy2_symbolic_ik_amd import SymbolicIK, ControlIK, DualArmIK
import bench
n = 1000
g = torch.Generator().manual_seed(1)
poses = torch.zeros((n, 2, 3), dtype=torch.float64)
poses[:, 0] = torch.tensor([0.35, -0.2, -0.25]) + 0.1 * (torch.rand((n, 3), generator=g, dtype=torch.float64) - 0.5)
poses[:, 1] = torch.tensor([0.0, -1.57, 0.0]) + 0.3 * (torch.rand((n, 3), generator=g, dtype=torch.float64) - 0.5)
poses = poses.cuda()
ik = SymbolicIK("r_arm")
res = ik.solve_batch(poses)
print({k: tuple(v.shape) for k, v in res.items()})
ctrl = ControlIK(urdf_path="config_files/reachy2_ik_minimal.urdf"); ctrl.nb_search_points = 64
traj = bench.make_config5_trajectories(64, 40, seed=3)                    # [n_steps, 12, n]
M = torch.zeros((64, 4, 4), dtype=torch.float64, device="cuda"); M[:, 3, 3] = 1
M[:, :3, :3] = traj[0, :9].T.reshape(64, 3, 3); M[:, :3, 3] = traj[0, 9:12].T
res = ctrl.symbolic_inverse_kinematics_batch("r_arm", M)
print("discrete", float(res["reachable"].float().mean()))
arm_ids = (torch.arange(64) % 2).to(torch.uint8).cuda()
res = ctrl.symbolic_inverse_kinematics_batch(arm_ids, M)
st = ctrl.new_continuous_state("r_arm", 64)
M_steps = torch.zeros((40, 64, 4, 4), dtype=torch.float64, device="cuda"); M_steps[..., 3, 3] = 1
M_steps[..., :3, :3] = traj[:, :9].permute(0, 2, 1).reshape(40, 64, 3, 3); M_steps[..., :3, 3] = traj[:, 9:12].permute(0, 2, 1)
res = ctrl.run_continuous_trajectories("r_arm", M_steps, st)
print("continuous", tuple(res["joints"].shape))
st2 = ctrl.new_continuous_state("r_arm", 64)
graph, out = ctrl.capture_continuous_trajectories("r_arm", traj, st2)
graph.replay(); torch.cuda.synchronize()
print("replayed", bool(torch.isfinite(out["joints"]).any()))
dual = DualArmIK(); res = dual.solve_batch((torch.arange(n) % 2).to(torch.uint8).cuda(), poses)
res1 = ik.solve_batch(poses)
p, R = ik.forward_kinematics_batch(res1["joints"])
err = ik.fk_residual_batch(poses, res1["joints"])
ok = res1["reachable"].bool()
print("fk residual of reachable poses", float(err[ok].abs().max()) if ok.any() else None)
