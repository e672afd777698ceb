#!/usr/bin/env python3
"""The words that tie the streams of rsik_control_continuous_run carry a 32-bit run number and only ever grow; before it wraps the
library drains what it has issued and starts the words over (RSIK_EDGE_SEQ_WRAP in rsik_lib.hip: 0xfffffff0 runs).  A test build puts
that point at run 5 (scripts/build_variant.py seqwrap -DRSIK_EDGE_SEQ_WRAP=5); this script issues 14 runs that continue one eventful
trajectory batch — overlapping (RSIK_OPT_CONT_GOALS_RESIDENT), two output sets in turn, 13 blocks through eight workspace slots —
and saves every run's outputs and the trajectory state:

    seq_wrap_check.py <librsik_hip.so | -> <out.pt>

tests/test_gpu_overlap.py runs it with the product library and with the test build and compares the files bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reachy2_symbolic_ik_amd import _abi  # noqa: E402

if sys.argv[1] != "-":
    _abi.use_library(os.path.abspath(sys.argv[1]))
import bench  # noqa: E402
from reachy2_symbolic_ik_amd import ControlIK  # noqa: E402

n_traj, n_steps, K = 300, 208, 14
whole = bench.make_config5_trajectories(n_traj, n_steps * K, seed=77, device=0)
sel = torch.arange(0, n_traj, 5, device=whole.device)
whole[n_steps * 3 + 40:, 9, sel] -= 0.25   # a jump of the goal: the continuity check trips, those trajectories stay latched
whole[n_steps * 3 + 40:, 11, sel] += 0.2
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
ctrl._solver.set_option(_abi.OPT_CONT_BLOCK_STEPS, 16)
st = ctrl.new_continuous_state("r_arm", n_traj)
outs = [None, None]
saved, forms = [], []
for k in range(K):
    goals = whole[k * n_steps:(k + 1) * n_steps].contiguous()
    outs[k & 1] = ctrl.run_continuous_trajectories("r_arm", goals, st, first_step_timed_out=(k == 0), current_pose=whole[0],
                                                   out=outs[k & 1], goals_resident=True)
    forms.append(int(outs[k & 1].run_form))
    if k >= 1:  # (the other set is the next run's: reading this one between the calls is within the promise)
        torch.cuda.synchronize()
    saved.append({key: v.cpu().clone() for key, v in outs[k & 1].items()})
    saved[-1]["cont_state"] = st.cpu().clone()
torch.cuda.synchronize()
ctrl._solver.synchronize()
torch.save({"runs": saved, "forms": forms}, sys.argv[2])
print("forms", forms, "latched", int((st[9] != 0).sum()))
