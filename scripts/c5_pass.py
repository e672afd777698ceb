#!/usr/bin/env python3
"""Six 4096 x 1000 passes of the continuous pipeline (config 5) and nothing else: the command rocprofv3 traces for
scripts/c5_timeline.py.  usage: c5_pass.py [alternative librsik_hip.so | -] [steps per block] [run mode: 1 phased (default), 3 single launch]
(steps per block = 1000: one block, the four phases one after the other, i.e. each kernel's time with the chip to itself)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reachy2_symbolic_ik_amd import ControlIK, _abi
blk = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _abi.use_library(os.path.abspath(sys.argv[1]))
n, n_steps = 4096, 1000
traj = bench.make_config5_trajectories(n, n_steps, seed=20250204, device=0)
if os.environ.get("C5_COHERENT"):  # every trajectory a copy of one of them: all lanes of a wave take the same path (a bound for what
    # making the waves coherent could win; C5_COHERENT = which trajectory)
    traj = traj[:, :, int(os.environ["C5_COHERENT"]):int(os.environ["C5_COHERENT"]) + 1].expand(-1, -1, n).contiguous()
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
cont0 = ctrl.new_continuous_state("r_arm", n)
ctrl._solver.set_option(_abi.OPT_CONT_BLOCK_STEPS, blk)
ctrl._solver.set_option(_abi.OPT_CONT_RUN_MODE, int(sys.argv[3]) if len(sys.argv) > 3 else _abi.CONT_RUN_PHASED)
out = {"joints": torch.empty((n_steps, n, 7), dtype=torch.float64, device="cuda"),
       "reachable": torch.empty((n_steps, n), dtype=torch.uint8, device="cuda"),
       "state": torch.empty((n_steps, n), dtype=torch.uint8, device="cuda")}
cont = cont0.clone()
if os.environ.get("C5_LATCHED"):  # the passes begin with most trajectories' emergency stop latched (two untraceable passes get there)
    ctrl.run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=out)
    ctrl.run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=False, current_pose=traj[0], out=out)
    torch.cuda.synchronize()
    cont0 = cont.clone()
    print("latched at the start of the passes:", int((cont0[9] != 0).sum()), "of", n, file=sys.stderr)
for _ in range(6):
    cont.copy_(cont0)
    ctrl.run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=not os.environ.get("C5_LATCHED"), current_pose=traj[0], out=out)
torch.cuda.synchronize()
