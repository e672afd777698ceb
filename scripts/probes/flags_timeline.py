#!/usr/bin/env python3
"""In-kernel timeline (first start / last end per kernel and block, 100 MHz clock) of one pass of rsik_control_continuous_run in a given
run mode, from a -DRSIK_PIPE_TIMING build.  usage: flags_timeline.py build/variants/pipe_timing.so [mode=4] [block steps=0]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reachy2_symbolic_ik_amd import _abi
_abi.use_library(os.path.abspath(sys.argv[1]))
import bench
from reachy2_symbolic_ik_amd import ControlIK
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 4
blk = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n, n_steps = 4096, 1000
traj = bench.make_config5_trajectories(n, n_steps, seed=20250204, device=0)
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
ctrl._solver.set_option(_abi.OPT_CONT_RUN_MODE, mode)
ctrl._solver.set_option(_abi.OPT_CONT_BLOCK_STEPS, blk)
cont0 = ctrl.new_continuous_state("r_arm", n)
out = None
cont = cont0.clone()
def one():
    global out
    cont.copy_(cont0)
    out = ctrl.run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=out)
for _ in range(6): one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): one()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per pass (eager, stamps on)", file=sys.stderr)
os.environ["RSIK_PIPE_TIMING_PRINT"] = "1"
one()   # prints the stamps of the pass before it
torch.cuda.synchronize()
