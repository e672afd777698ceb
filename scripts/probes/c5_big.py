#!/usr/bin/env python3
"""Large trajectory batches through the continuous pipeline against the step kernel (debug probe): n trajectories x steps."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from reachy2_symbolic_ik_amd import ControlIK, _abi
n, n_steps = int(sys.argv[1]), int(sys.argv[2])
traj = bench.make_config5_trajectories(n, n_steps, seed=7, device=0)
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
res = {}
for mode in (_abi.CONT_RUN_PHASED, _abi.CONT_RUN_STEPS):
    ctrl._solver.set_option(_abi.OPT_CONT_RUN_MODE, mode)
    cont = ctrl.new_continuous_state("r_arm", n)
    cont0 = cont.clone()
    out = None
    for rep in range(3):
        cont.copy_(cont0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = ctrl.run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=out)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    res[mode] = (out, cont.clone())
    print(f"mode {mode}: {ms:.3f} ms, {n * n_steps / ms / 1e6:.2f} G steps/s", flush=True)
a, b = res[_abi.CONT_RUN_PHASED], res[_abi.CONT_RUN_STEPS]
print("flags equal", torch.equal(a[0]["reachable"], b[0]["reachable"]), torch.equal(a[0]["state"], b[0]["state"]),
      "max joint diff", float((a[0]["joints"] - b[0]["joints"]).abs().max()), "theta equal", torch.equal(a[1][0], b[1][0]))
