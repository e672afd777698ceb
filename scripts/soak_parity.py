#!/usr/bin/env python3
"""Large seeded parity soak (not part of the test suite): the fused solve on mixed r / l batches of uniformly random
poses (all outcomes) and on reachable-only batches with random theta fractions, against the CPU checker.

    python scripts/soak_parity.py [n_per_seed] [n_seeds]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from reachy2_symbolic_ik_amd import DualArmIK  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dual = bench._quiet(DualArmIK)
ar, al = orc.Arm("r_arm", 0.03), orc.Arm("l_arm", 0.03)
nt = max(1, os.cpu_count() or 1)
worst, flips, total = 0.0, 0, 0
for seed in range(seeds):
    rng = np.random.default_rng(1000 + seed)
    arm = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    pos = np.stack([rng.uniform(-0.7, 0.7, n), np.where(arm == 1, 0.2, -0.2) + rng.uniform(-0.7, 0.7, n), rng.uniform(-0.7, 0.7, n)], axis=1)
    eul = rng.uniform(-np.pi, np.pi, size=(n, 3))
    frac = rng.uniform(size=n)
    soa = torch.as_tensor(np.ascontiguousarray(np.concatenate([pos.T, eul.T], axis=0))).cuda()
    for policy, theta in ((0, "interval0"), (2, ("fraction", torch.as_tensor(frac).cuda()))):
        res = {k: v.cpu().numpy() for k, v in dual.solve_batch(torch.as_tensor(arm).cuda(), soa, theta=theta).items()}
        ref = orc.solve_batch(ar, al, pos, eul, arm_id=arm, theta_policy=policy, theta_in=frac if policy else None, nthreads=nt)
        bad = (res["reachable"] != ref["reachable"]) | (res["state"] != ref["state"])
        m = (ref["reachable"] == 1) & ~bad & (np.abs(ref["joints"][:, 3]) > 1e-9)   # (fully extended arm: j2 + j6 only)
        err = max(np.max(np.abs(res["joints"][m] - ref["joints"][m])), np.max(np.abs(res["interval"][m] - ref["interval"][m])),
                  np.max(np.abs(res["elbow"][m] - ref["elbow"][m])))
        worst, flips, total = max(worst, float(err)), flips + int(bad.sum()), total + n
        print(f"seed {seed} policy {policy}: {n} poses, {int(ref['reachable'].sum())} reachable, flag/state mismatches {int(bad.sum())}, "
              f"max |error| {err:.2e}", flush=True)
print(f"TOTAL {total} poses: {flips} flag/state mismatches, worst error {worst:.2e} rad / m")

# ControlIK discrete mode: random goal matrices (all outcomes), both arms in one launch, two grid sizes
from reachy2_symbolic_ik_amd import ControlIK  # noqa: E402
from reachy2_symbolic_ik_amd.constants import euler_xyz_extrinsic  # noqa: E402,F401

ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF)
cr, cl = orc.Arm("r_arm", -1.01), orc.Arm("l_arm", -1.01)
nd = min(n, 1 << 20)
flips_d, worst_d, total_d = 0, 0.0, 0
for seed in range(min(seeds, 3)):
    rng = np.random.default_rng(2000 + seed)
    arm = (rng.uniform(size=nd) < 0.5).astype(np.uint8)
    # half of the positions in the well-reachable region so that the theta search really runs
    centre = np.where(rng.uniform(size=(nd, 1)) < 0.5, np.array([[0.35, 0.0, -0.15]]), np.zeros((1, 3)))
    pos = centre + np.stack([rng.uniform(-0.5, 0.5, nd), np.where(arm == 1, 0.2, -0.2) + rng.uniform(-0.5, 0.5, nd), rng.uniform(-0.5, 0.5, nd)], axis=1)
    e = rng.uniform(-np.pi, np.pi, size=(nd, 3))
    ca, sa, cb, sb, cc, sc = np.cos(e[:, 0]), np.sin(e[:, 0]), np.cos(e[:, 1]), np.sin(e[:, 1]), np.cos(e[:, 2]), np.sin(e[:, 2])
    M = np.zeros((nd, 4, 4))
    M[:, 0, 0] = cc * cb; M[:, 0, 1] = cc * sb * sa - sc * ca; M[:, 0, 2] = cc * sb * ca + sc * sa
    M[:, 1, 0] = sc * cb; M[:, 1, 1] = sc * sb * sa + cc * ca; M[:, 1, 2] = sc * sb * ca - cc * sa
    M[:, 2, 0] = -sb; M[:, 2, 1] = cb * sa; M[:, 2, 2] = cb * ca
    M[:, :3, 3] = pos
    M[:, 3, 3] = 1.0
    for nb in (20, 64):
        ctrl.nb_search_points = nb
        res = {k: v.cpu().numpy() for k, v in ctrl.symbolic_inverse_kinematics_batch(torch.as_tensor(arm).cuda(), M).items()}
        ref = orc.control_discrete_batch(cr, cl, M, arm_id=arm, nb_search_points=nb, nthreads=nt)
        bad = (res["reachable"] != ref["reachable"]) | (res["state"] != ref["state"])
        err = float(np.max(np.abs(res["joints"][~bad] - ref["joints"][~bad])))
        flips_d, worst_d, total_d = flips_d + int(bad.sum()), max(worst_d, err), total_d + nd
        print(f"discrete seed {seed} nb {nb}: {nd} matrices, found {int(ref['reachable'].sum())}, states {np.bincount(ref['state'], minlength=7).tolist()}, "
              f"flag/state mismatches {int(bad.sum())}, max |joint error| {err:.2e}", flush=True)
print(f"DISCRETE TOTAL {total_d} matrices: {flips_d} flag/state mismatches, worst error {worst_d:.2e} rad")
