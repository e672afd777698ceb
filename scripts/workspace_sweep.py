#!/usr/bin/env python3
"""Batched equivalent of the reference's workspace sweep (src/benchmark/ik_comparison.py:137-181 `task_space_test`):
a regular grid of goal positions inside a sphere around the shoulder (x >= 0), every 45 degree roll/pitch/yaw
combination at each, one launch of the fused solve kernel, and an on-device FK check of every solution.

    python scripts/workspace_sweep.py [--step 0.15] [--angle-step 45] [--arm r_arm]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reachy2_symbolic_ik_amd import SymbolicIK  # noqa: E402
from reachy2_symbolic_ik_amd.constants import STATE_STRINGS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--step", type=float, default=0.15)
    ap.add_argument("--angle-step", type=int, default=45)
    ap.add_argument("--arm", default="r_arm")
    ap.add_argument("--arm-length", type=float, default=0.5, help="radius of the swept sphere (the reference uses 0.5)")
    args = ap.parse_args()
    ik = SymbolicIK(args.arm)
    s = np.asarray(ik.shoulder_position, dtype=np.float64)
    L = args.arm_length
    ax = [np.arange(s[k] - L, s[k] + L + args.step, args.step) for k in range(3)]
    P = np.stack(np.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3)
    P = P[(np.linalg.norm(P - s, axis=1) <= L) & (P[:, 0] >= 0)]
    ang = np.radians(np.arange(0, 360, args.angle_step))
    E = np.stack(np.meshgrid(ang, ang, ang, indexing="ij"), -1).reshape(-1, 3)
    pos = np.repeat(P, len(E), axis=0)
    eul = np.tile(E, (len(P), 1))
    soa = torch.as_tensor(np.ascontiguousarray(np.concatenate([pos.T, eul.T], axis=0))).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = ik.solve_batch(soa)
    err = ik.fk_residual_batch(soa, res["joints"])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = res["reachable"].bool()
    print("TASK SPACE TEST (%s)" % args.arm)
    print("goal_poses : ", len(pos))
    print("time : %.6f s (solve + FK check, one launch each)" % dt)
    print("reachable poses : ", int(ok.sum()))
    print("total poses : ", len(pos))
    codes, counts = torch.unique(res["state"], return_counts=True)
    for c, k in zip(codes.tolist(), counts.tolist()):
        print("  %-40s %d" % (STATE_STRINGS[c], k))
    e = err[ok]
    exact = (e[:, 0] < 1e-9) & (e[:, 1] < 1e-9)
    print("FK(IK(pose)) == pose to 1e-9 for %d of %d reachable poses; the other %d goals were moved by the solver "
          "(backward shift, min-distance reduce, elbow projection): max shift %.4f m" %
          (int(exact.sum()), int(ok.sum()), int((~exact).sum()), float(e[:, 0].max()) if len(e) else 0.0))


if __name__ == "__main__":
    main()
