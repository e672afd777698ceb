#!/usr/bin/env python3
"""Config 5 (4096 trajectories x 1000 control steps per pass) under the pipeline's tuning option
(RSIK_OPT_CONT_BLOCK_STEPS: size of a run's first block): ms per pass and steps/s, the results of every variant compared
bit for bit with the first one.  usage: c5_ab.py [passes]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from reachy2_symbolic_ik_amd import ControlIK, _abi  # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n, n_steps = 4096, 1000
traj = bench.make_config5_trajectories(n, n_steps, seed=20250204, device=0)
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
cont0 = ctrl.new_continuous_state("r_arm", n)
ref = None
for blk in [int(b) for b in os.environ.get("C5_BLOCKS", "0,64,128,192,256,352,512,1000,0").split(",")]:
    ctrl._solver.set_option(_abi.OPT_CONT_BLOCK_STEPS, blk)
    out = {"joints": torch.empty((n_steps, n, 7), dtype=torch.float64, device="cuda"),
           "reachable": torch.empty((n_steps, n), dtype=torch.uint8, device="cuda"),
           "state": torch.empty((n_steps, n), dtype=torch.uint8, device="cuda")}
    cont = cont0.clone()

    def one_pass():
        cont.copy_(cont0)
        ctrl.run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=out)

    for _ in range(3):
        one_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        one_pass()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / passes * 1e3
    if ref is None:
        ref = {k: v.clone() for k, v in out.items()}
        ref["cont"] = cont.clone()
        same = "reference"
    else:
        same = all(torch.equal(ref[k].view(torch.uint8), out[k].view(torch.uint8)) for k in out) and torch.equal(ref["cont"].view(torch.uint8), cont.view(torch.uint8))
        same = "bit-identical" if same else "DIFFERENT"
    print(f"block {blk:5d} steps: {ms:8.3f} ms per pass  {n * n_steps / ms / 1e6:7.2f} G steps/s  [{same}]", flush=True)
