#!/usr/bin/env python3
"""Config 5 pass issued eagerly vs replayed from a hipGraph that captured the whole four-stream pipeline of one pass
(the side streams join the capture through the events the run records): ms per pass and bit-identity of the results."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
def timed(f, reps=20):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best
for _ in range(5): one()
torch.cuda.synchronize()
ref = out["joints"].clone(); ref_state = cont.clone()
print(f"eager: {timed(one):.3f} ms per pass")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2): one()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        one()
out["joints"].zero_()
g.replay(); torch.cuda.synchronize()
print("graph replay bit-identical:", bool(torch.equal(out["joints"].view(torch.uint8), ref.view(torch.uint8))) and bool(torch.equal(cont.view(torch.uint8), ref_state.view(torch.uint8))))
print(f"graph: {timed(g.replay):.3f} ms per pass")
