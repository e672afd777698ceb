#!/usr/bin/env python3
"""Round 6: a run that CONTINUES a trajectory batch (no re-initialisation, previous_sol as the run before left it — 8 % of config 5's
steps have a joint beyond +-pi) with and without the joints phase's turn hint (RSIK_OPT_CONT_PHASED_VARIANT bit 64 = without).
    python scripts/probes/c5_continuing_run.py [n_traj] [steps per run] [runs]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from reachy2_symbolic_ik_amd import ControlIK, _abi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
R = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
hs = ctrl._solver
traj = bench.make_config5_trajectories(n, T * R, seed=20250204, device=0)
pieces = [traj[k * T:(k + 1) * T].contiguous() for k in range(R)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ref = None
for variant in (64, 0, 64, 0):
    hs.set_option(_abi.OPT_CONT_PHASED_VARIANT, variant)
    st = ctrl.new_continuous_state("r_arm", n)
    outs = [None, None]
    ms = []
    for k in range(R):
        torch.cuda.synchronize()
        e0.record()
        outs[k & 1] = ctrl.run_continuous_trajectories("r_arm", pieces[k], st, first_step_timed_out=(k == 0), current_pose=traj[0], out=outs[k & 1])
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    j = outs[(R - 1) & 1]["joints"]
    wound = float((j.abs() > 3.141592653589793).any(dim=2).float().mean())
    got = (outs[(R - 1) & 1]["state"].clone(), outs[(R - 1) & 1]["reachable"].clone(), st[0].clone(), j.clone())
    same = "reference" if ref is None else ("flags / states / theta identical, max joint diff %.1e" % float((got[3] - ref[3]).abs().max())
                                            if all(torch.equal(a, b) for a, b in zip(got[:3], ref[:3])) else "FLAGS DIFFER")
    ref = ref or got
    print(f"variant {variant:2d} ({'no hint' if variant else 'hint   '}): ms per run " + " ".join(f"{v:.3f}" for v in ms) + f" | last run: a joint beyond pi in {wound:.3f} of the steps [{same}]", flush=True)
hs.set_option(_abi.OPT_CONT_PHASED_VARIANT, 0)
