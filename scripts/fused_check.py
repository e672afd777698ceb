#!/usr/bin/env python3
"""The single-launch form of rsik_control_continuous_run (RSIK_CONT_RUN_FUSED) against the phased pipeline and the step
kernel: same bits (flags, state codes, carried theta, latch rows; joints bit-identical to the phased pipeline — it is the
same device code — and to 1e-9 against the step kernel), then its time per pass launched eagerly, and a summary of the
in-kernel work-item trace (RSIK_OPT_CONT_TRACE).
usage: fused_check.py [trajectories] [steps] [passes]   (environment: FUSED_S, FUSED_L, FUSED_SP = block / look-ahead / prepare steps)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from reachy2_symbolic_ik_amd import ControlIK, _abi as A  # noqa: E402

n_traj = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
hs = ctrl._solver
for opt, env in ((A.OPT_CONT_BLOCK_STEPS, "FUSED_S"), (A.OPT_CONT_LOOKAHEAD, "FUSED_L"), (A.OPT_CONT_PREP_STEPS, "FUSED_SP"),
                 (A.OPT_CONT_CHAIN_LAG, "FUSED_CL"), (A.OPT_CONT_JOINT_GROUPS, "FUSED_JH")):
    if os.environ.get(env):
        hs.set_option(opt, int(os.environ[env]))
traj = bench.make_config5_trajectories(n_traj, n_steps, device=0)
cont0 = ctrl.new_continuous_state("r_arm", n_traj)


def run(mode, out=None):
    hs.set_option(A.OPT_CONT_RUN_MODE, mode)
    st = cont0.clone()
    o = ctrl.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0], out=out)
    hs.synchronize()
    return o, st


res = {}
blk = hs.get_option(A.OPT_CONT_BLOCK_STEPS)
for name, mode in (("steps", A.CONT_RUN_STEPS), ("phased", A.CONT_RUN_PHASED), ("fused", A.CONT_RUN_FUSED), ("flags", A.CONT_RUN_FLAGS)):
    if name == "steps" and n_traj * n_steps > 600_000:
        continue
    hs.set_option(A.OPT_CONT_BLOCK_STEPS, blk if name in ("fused", "flags") else 0)
    o, st = run(mode)
    res[name] = ({k: v.clone() for k, v in o.items()}, st.clone())
hs.set_option(A.OPT_CONT_BLOCK_STEPS, blk)
bad = 0
for mine, other in (("fused", "phased"), ("fused", "steps"), ("flags", "phased"), ("flags", "steps")):
    if other not in res:
        continue
    (a, sa), (b, sb) = res[mine], res[other]
    same = all(torch.equal(a[k], b[k]) for k in ("reachable", "state"))
    same = same and torch.equal(sa[0].view(torch.uint8), sb[0].view(torch.uint8)) and torch.equal(sa[8:12], sb[8:12])
    fin = torch.isfinite(a["joints"]) & torch.isfinite(b["joints"])
    nan_same = torch.equal(torch.isfinite(a["joints"]), torch.isfinite(b["joints"]))
    worst = float((a["joints"][fin] - b["joints"][fin]).abs().max())
    bits = torch.equal(a["joints"].view(torch.int64), b["joints"].view(torch.int64)) and torch.equal(sa.view(torch.int64), sb.view(torch.int64))
    ok = same and nan_same and worst <= (0.0 if other == "phased" else 1e-9)
    print(f"{mine} vs {other}: flags/states/theta/latch {'identical' if same else 'DIFFER'}, joints max diff {worst:.3e}, "
          f"every bit identical: {bits}  -> {'OK' if ok else 'MISMATCH'}")
    bad += 0 if ok else 1

# ---- time per pass, eager
out = {k: torch.empty_like(v) for k, v in res["fused"][0].items()}
for name, mode in (("phased", A.CONT_RUN_PHASED), ("fused", A.CONT_RUN_FUSED), ("flags", A.CONT_RUN_FLAGS)):
    hs.set_option(A.OPT_CONT_RUN_MODE, mode)
    hs.set_option(A.OPT_CONT_BLOCK_STEPS, blk if name in ("fused", "flags") else 0)
    st = cont0.clone()

    def one():
        st.copy_(cont0)
        ctrl.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0], out=out)

    for _ in range(5):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        one()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / passes * 1e3
    print(f"{name:7s} eager: {ms:.4f} ms per pass, {n_traj * n_steps / ms / 1e6:.2f} G steps/s")
hs.set_option(A.OPT_CONT_BLOCK_STEPS, blk)

# ---- the work-item trace of one pass of the single launch (FUSED_TRACE_MODE=4: the flag-synchronised form's persistent kernels)
tmode = int(os.environ.get("FUSED_TRACE_MODE", A.CONT_RUN_FUSED))
hs.set_option(A.OPT_CONT_RUN_MODE, tmode)
hs.set_option(A.OPT_CONT_TRACE, 64)
run(tmode, out)
run(tmode, out)
rec = hs.control_continuous_trace()
hs.set_option(A.OPT_CONT_TRACE, 0)
if len(rec):
    t_claim, t_ready, t_end, what = (rec[:, k].astype(np.int64) for k in range(4))
    kind = what & 0xF
    blkno = (what >> 4) & 0xFFFF
    base = t_claim.min()
    us = lambda x: (x - base) / 100.0  # noqa: E731
    print(f"trace: {len(rec)} records, span {us(t_end.max()):.1f} us")
    for kk, nm in ((2, "prepare"), (1, "joints"), (0, "chain"), (3, "theta block")):
        m = kind == kk
        if not m.any():
            continue
        wait = (t_ready - t_claim)[m] / 100.0
        work = (t_end - t_ready)[m] / 100.0
        print(f"  {nm:11s} {int(m.sum()):6d} items: first claimed {us(t_claim[m].min()):7.1f}, last done {us(t_end[m].max()):7.1f} us; "
              f"waited mean {wait.mean():6.2f} max {wait.max():7.2f}; worked mean {work.mean():6.2f} max {work.max():7.2f} us")
    nb = int(blkno.max()) + 1
    for b in sorted(set([0, 1, 2, nb // 2, nb - 2, nb - 1])):
        if b < 0:
            continue
        row = []
        for kk, nm in ((2, "P"), (3, "T"), (1, "J"), (0, "C")):
            m = (kind == kk) & (blkno == b)
            if m.any():
                row.append(f"{nm} {us(t_ready[m].min()):6.1f}-{us(t_end[m].max()):6.1f}")
        print(f"  block {b:3d}: " + "  ".join(row))
    if os.environ.get("FUSED_TRACE_OUT"):
        np.save(os.environ["FUSED_TRACE_OUT"], rec)
hs.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_AUTO)
print("TOTAL", "0 mismatches" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
