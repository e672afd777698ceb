#!/usr/bin/env python3
"""Round 6: many overlapping passes (RSIK_OPT_CONT_GOALS_RESIDENT), every pass checked — a missing dependency between the streams of
consecutive runs would show as a rare pass whose outputs differ.

    python scripts/probes/c5_overlap_stress.py [passes, default 20000] [n_traj 4096] [n_steps 1000]

The bench's config-5 protocol (every pass resets the trajectory state and re-initialises every trajectory, same goals, two sets of
output buffers in turn).  Behind every pass three numbers are reduced on the device from the set it wrote — the sum of its finite
joints, of its state codes and of its reachable flags — and kept; at the end every pass's triple must equal the first pass's exactly
(the passes compute the same thing, bit for bit), and so must the trajectory state.  The reductions read the set the NEXT pass does
not write: within the promise.  Every 1000 passes the host waits (rsik_sync) and prints a line.  C5_BLOCKS = RSIK_OPT_CONT_BLOCK_STEPS,
C5_EVENTFUL=1: goals that jump (latches, step-by-step chunks, emergency rows written over the prepare phase's)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from reachy2_symbolic_ik_amd import ControlIK, _abi  # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
dev = torch.device("cuda", 0)
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
ctrl._solver.set_option(_abi.OPT_CONT_BLOCK_STEPS, int(os.environ.get("C5_BLOCKS", "0")))  # (16 with 208 steps: 13 blocks through 8 slots)
traj = bench.make_config5_trajectories(n, n_steps, seed=20250204, device=0)
if os.environ.get("C5_EVENTFUL"):  # every fifth trajectory's goal jumps part-way: the continuity check trips, the trajectory stays latched
    sel = torch.arange(0, n, 5, device=traj.device)
    traj[int(n_steps * 0.4):, 9, sel] -= 0.25
    traj[int(n_steps * 0.4):, 11, sel] += 0.2
cont0 = ctrl.new_continuous_state("r_arm", n)
cont = cont0.clone()
outs = [None, None]
sums = torch.zeros((passes, 3), dtype=torch.float64, device=dev)
state_sum = torch.zeros((passes,), dtype=torch.float64, device=dev)
forms = {}
t0 = time.perf_counter()
for k in range(passes):
    cont.copy_(cont0)
    o = ctrl.run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=outs[k & 1], goals_resident=True)
    outs[k & 1] = o
    forms[o.run_form_name] = forms.get(o.run_form_name, 0) + 1
    sums[k, 0] = o["joints"].nan_to_num(0.0).sum()
    sums[k, 1] = o["state"].sum(dtype=torch.float64)
    sums[k, 2] = o["reachable"].sum(dtype=torch.float64)
    state_sum[k] = cont.nan_to_num(0.0).sum()
    if (k + 1) % 1000 == 0:
        ctrl._solver.synchronize()
        bad_so_far = int(((sums[: k + 1] != sums[0]).any(dim=1) | (state_sum[: k + 1] != state_sum[0])).sum())
        print(f"{k + 1} passes, {time.perf_counter() - t0:.1f} s, passes that differ from the first so far: {bad_so_far}", flush=True)
torch.cuda.synchronize()
bad = ((sums != sums[0]).any(dim=1) | (state_sum != state_sum[0])).nonzero().flatten().tolist()
print(f"{passes} passes of {n} x {n_steps} steps, run forms {forms}; checksums of pass 0: joints {float(sums[0, 0]):.12e}, states {int(sums[0, 1])}, "
      f"reachable {int(sums[0, 2])}, trajectory state {float(state_sum[0]):.12e}")
print("TOTAL", "every pass identical to the first" if not bad else f"{len(bad)} passes differ: {bad[:20]}")
sys.exit(1 if bad else 0)
