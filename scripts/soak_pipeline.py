#!/usr/bin/env python3
"""Soak of the trajectory pipeline's sequential phases: random jumpy trajectories (continuity trips, +-6 pi limits reached by
winding joints, unreachable stretches, exact repeats = "stay" steps) through rsik_control_continuous_run as the phased
pipeline and as one launch of the step kernel per control step, for both arms x both constrained modes x two rate
limits: flags, state codes, the carried theta, the latch / init rows and the emergency cause bits must be the same bits,
joints and previous_sol equal to 1e-9 (the pipeline writes a quiet step as raw joint + whole turns instead of previous +
angle_diff(raw, previous): the last bits differ, amplified where the arm is stretched out); and — round 6 — as three
overlapping runs (RSIK_OPT_CONT_GOALS_RESIDENT), which must be the pipeline's bits exactly.
usage: soak_pipeline.py [trajectories] [steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from reachy2_symbolic_ik_amd import ControlIK, _abi  # noqa: E402

n_traj = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 203
dev = torch.device("cuda", 0)


def trajectories(seed, arm):
    """[n_steps, 12, n_traj]: smooth sinusoids like config 5, plus per-trajectory events: a jump of the goal (a third of
    them), a long winding of the wrist roll (a third), goals far out of reach for a stretch (a sixth)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64).to(dev)  # noqa: E731
    k = torch.arange(n_steps, dtype=torch.float64, device=dev)[:, None]
    t = k / 60.0 + 11.0 + r(n_traj)[None, :] * 40.0
    c0 = [0.45, -0.2 if arm == "r_arm" else 0.2, -0.1, 0.0, -np.pi / 2, 0.0]
    amp = [0.3, 0.3, 0.3, np.pi / 4, np.pi / 4, np.pi / 4]
    freq = [0.6, 0.34, 0.78, 0.18, 0.31, 0.47]
    v = [c + a * torch.sin(f * t) for c, a, f in zip(c0, amp, freq)]
    kind = (r(n_traj) * 6).floor()
    at = (r(n_traj) * (n_steps - 20) + 10).floor()
    after = (k >= at[None, :]).double()
    v[0] = v[0] + after * (kind < 2).double()[None, :] * 0.25 * (r(n_traj)[None, :] - 0.5)       # a jump of the goal
    v[5] = v[5] + (kind == 2).double()[None, :] * k * 0.12 + (kind == 3).double()[None, :] * k * -0.2  # winding wrist
    far = ((k >= at[None, :]) & (k < at[None, :] + 25)).double() * (kind == 4).double()[None, :]
    v[0] = v[0] + far * 2.0                                                                            # out of reach
    hold = ((k >= at[None, :]) & (k < at[None, :] + 6)).double() * (kind == 5).double()[None, :]
    idx = torch.where(hold.bool(), at[None, :].expand(n_steps, -1), k.expand(-1, n_traj)).long()
    v = [torch.gather(x, 0, idx) for x in v]                                                           # exact repeats
    ca, sa, cb, sb, cc, sc = torch.cos(v[3]), torch.sin(v[3]), torch.cos(v[4]), torch.sin(v[4]), torch.cos(v[5]), torch.sin(v[5])
    rows = [cc * cb, cc * sb * sa - sc * ca, cc * sb * ca + sc * sa, sc * cb, sc * sb * sa + cc * ca, sc * sb * ca - cc * sa,
            -sb, cb * sa, cb * ca, v[0], v[1], v[2]]
    return torch.stack(rows, dim=1).contiguous()


ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
A = _abi
bad = 0
for ai, arm in enumerate(("r_arm", "l_arm")):
    for mode in ("unconstrained", "low_elbow"):
        for dmax in (0.01, 0.3):
            traj = trajectories(100 * ai + len(mode) + int(dmax * 100), arm)
            res = {}
            for name, run_mode in (("steps", A.CONT_RUN_STEPS), ("pipeline", A.CONT_RUN_PHASED)):
                ctrl._solver.set_option(A.OPT_CONT_RUN_MODE, run_mode)
                st = ctrl.new_continuous_state(arm, n_traj)
                out = ctrl.run_continuous_trajectories(arm, traj, st, first_step_timed_out=True, current_pose=traj[0],
                                                       constrained_mode=mode, d_theta_max=dmax)
                torch.cuda.synchronize()
                res[name] = {k: v.clone() for k, v in out.items()}
                res[name]["cont_state"] = st.clone()
            # round 6: the same run issued three times back to back with RSIK_OPT_CONT_GOALS_RESIDENT — the second and third overlap the
            # run before them (prepare phase beside its tail, slots taking turns, theta kernels waiting for their prepare kernels
            # themselves) — must give the pipeline's bits every time
            ctrl._solver.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_PHASED)
            outs = [None, None]
            forms = []
            for rep in range(3):
                st = ctrl.new_continuous_state(arm, n_traj)
                outs[rep & 1] = ctrl.run_continuous_trajectories(arm, traj, st, first_step_timed_out=True, current_pose=traj[0],
                                                                 constrained_mode=mode, d_theta_max=dmax, out=outs[rep & 1], goals_resident=True)
                forms.append(outs[rep & 1].run_form)
            torch.cuda.synchronize()
            over = {k: v for k, v in outs[0].items()}
            over["cont_state"] = st
            overlapped_same = forms[1:] == [A.CONT_FORM_PHASED_OVERLAPPED] * 2 and all(
                torch.equal(over[k].contiguous().view(torch.uint8), res["pipeline"][k].contiguous().view(torch.uint8)) for k in over)
            a, b = res["steps"], res["pipeline"]
            same = all(torch.equal(a[k], b[k]) for k in ("reachable", "state"))
            sa, sb = a["cont_state"], b["cont_state"]
            same = same and torch.equal(sa[0].view(torch.uint8), sb[0].view(torch.uint8)) and torch.equal(sa[8:12], sb[8:12])
            worst = max(float((a["joints"] - b["joints"]).abs().max()), float((sa[1:8] - sb[1:8]).abs().max()),
                        float((sa[12:19] - sb[12:19]).abs().max()))
            same = same and worst <= 1e-9 and overlapped_same
            st = res["steps"]["cont_state"]
            j = res["steps"]["joints"]
            print(f"{arm} {mode:13s} d_theta_max {dmax}: {n_traj} x {n_steps} steps, reachable {float(res['steps']['reachable'].float().mean()):.2f}, "
                  f"latched {int((st[9] != 0).sum())}, joints beyond pi in {float((j.abs() > np.pi).any(dim=2).float().mean()):.3f} of the steps, "
                  f"max |joint| {float(j[torch.isfinite(j)].abs().max()):.2f}: {'flags / states / theta / latch identical, joints to %.1e; overlapped runs bit-identical to the pipeline' % worst if same else 'MISMATCH (joints %.3e, overlapped runs %s)' % (worst, 'identical' if overlapped_same else 'DIFFER')}")
            bad += 0 if same else 1
ctrl._solver.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_AUTO)
print("TOTAL", "0 mismatches" if bad == 0 else f"{bad} configurations differ")
sys.exit(1 if bad else 0)
