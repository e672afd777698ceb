#!/usr/bin/env python3
"""Captures one rsik_control_continuous_run call into a hipGraph on a fresh side stream and replays it (debug probe)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from reachy2_symbolic_ik_amd import ControlIK
n, n_steps = 700, int(sys.argv[1]) if len(sys.argv) > 1 else 150
traj = bench.make_config5_trajectories(n, n_steps, seed=4242, device=0)
c = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
st0 = c.new_continuous_state("r_arm", n)
st = st0.clone()
def one():
    st.copy_(st0)
    c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0])
one(); torch.cuda.synchronize()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    one(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    print("capturing", flush=True)
    with torch.cuda.graph(g, stream=side):
        one()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed OK", flush=True)
