#!/usr/bin/env python3
"""A/B of a pipeline variant that the library toggles every 8 calls (debug builds): ms per pass of both halves, interleaved."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from reachy2_symbolic_ik_amd import ControlIK
n, n_steps = 4096, 1000
traj = bench.make_config5_trajectories(n, n_steps, seed=20250204, device=0)
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
cont0 = ctrl.new_continuous_state("r_arm", n)
out = {"joints": torch.empty((n_steps, n, 7), dtype=torch.float64, device="cuda"),
       "reachable": torch.empty((n_steps, n), dtype=torch.uint8, device="cuda"),
       "state": torch.empty((n_steps, n), dtype=torch.uint8, device="cuda")}
cont = cont0.clone()
def one():
    cont.copy_(cont0)
    ctrl.run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=out)
res = {0: [], 1: []}
for rep in range(16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): one()
    torch.cuda.synchronize(); res[rep % 2].append((time.perf_counter() - t0) / 8 * 1e3)
for k in (0, 1):
    v = sorted(res[k][1:])
    print(f"variant {k}: median {v[len(v)//2]:.4f} ms per pass, min {v[0]:.4f}  ({[round(x,4) for x in res[k]]})")
