#!/usr/bin/env python3
"""What the start-up search (cont_init_kernel, control_ik.py:296-325) costs a pass: config-5 passes with every trajectory (re)initialising on
the first step (the bench's protocol) against passes that carry on from an initialised state (first_step_timed_out = False)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from reachy2_symbolic_ik_amd import ControlIK
n, N = 4096, 1000
traj = bench.make_config5_trajectories(n, N, device=0)
c = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
st0 = c.new_continuous_state("r_arm", n)
st = st0.clone()
c.run_continuous_trajectories("r_arm", traj[:1].contiguous(), st, first_step_timed_out=True, current_pose=traj[0])
torch.cuda.synchronize()
st1 = st.clone()  # a state initialised at the trajectories' first goal (the state at their END would trip the continuity check)
out = c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0])
torch.cuda.synchronize()
for rep in range(3):
    for timed_out, base in ((True, st0), (False, st1)):
        def one():
            st.copy_(base)
            c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=timed_out, current_pose=traj[0], out=out)
        for _ in range(5): one()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): one()
        torch.cuda.synchronize()
        print(f"first_step_timed_out={timed_out}: {(time.perf_counter() - t0) / 20 * 1e3:.4f} ms per pass")
