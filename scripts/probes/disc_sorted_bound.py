#!/usr/bin/env python3
"""Config 3: what could a repartition of a workgroup's poses after `reach` / after the preferred-theta shortcut win at most?
The same 256 Ki goal matrices solved in their generated order and SORTED by the path they take through the kernel (shortcut hit /
grid search needed and found / nothing found), i.e. with every wave homogeneous — the perfect repartition, at no cost.  The path
of each pose comes from a -DRSIK_DISC_CLASS_PROBE build (a child process); the timings are the product library's.
usage: disc_sorted_bound.py [--no-build]"""
import json
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PROBE = os.path.join(ROOT, "build", "variants", "disc_class.so")
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from reachy2_symbolic_ik_amd import _abi
_abi.use_library(sys.argv[1])
import bench
from reachy2_symbolic_ik_amd import ControlIK
M = bench.make_config3_matrices(1 << 18)
c = bench._quiet(ControlIK, urdf_path=bench.URDF)
c.nb_search_points = 64
res = c.symbolic_inverse_kinematics_batch("r_arm", M)
np.save(sys.argv[2], res["emergency"].cpu().numpy())
'''
if "--no-build" not in sys.argv:
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "build_variant.py"), "disc_class", "-DRSIK_DISC_CLASS_PROBE"], check=True,
                   stdout=subprocess.DEVNULL)
tmp = os.path.join(ROOT, "gpurun_out", "disc_class.npy")
os.makedirs(os.path.dirname(tmp), exist_ok=True)
subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, PROBE, tmp], check=True)
cls = np.load(tmp)

import bench  # noqa: E402
from reachy2_symbolic_ik_amd import ControlIK  # noqa: E402
from reachy2_symbolic_ik_amd.control_ik import matrices_to_m12_soa  # noqa: E402

n = 1 << 18
M = bench.make_config3_matrices(n)
need, found, coop = (cls & 16) != 0, (cls & 32) != 0, (cls & 64) != 0
print(f"{n} poses: shortcut hit {np.mean(~need & found):.3f}, grid search needed {need.mean():.3f} (found {np.mean(need & found):.3f}, "
      f"nothing found {np.mean(need & ~found):.3f}; cooperative sweep {coop.mean():.3f}), not reachable at all {np.mean(~need & ~found):.3f}")
w = cls.reshape(-1, 64)
print(f"waves (generated order): with >= 1 lane needing the search {np.mean(((w & 16) != 0).any(axis=1)):.3f}, "
      f"with >= 1 lane computing joints {np.mean(((w & 32) != 0).any(axis=1)):.3f}")
c = bench._quiet(ControlIK, urdf_path=bench.URDF)
c.nb_search_points = 64
dev = torch.device("cuda", 0)
key = (need.astype(int) * 2 + found.astype(int))
# ("within workgroups": each block of 256 consecutive poses sorted on its own — what a repartition through LDS could do, every
# compute unit keeping its mix of work; the global sorts also move whole classes onto the same compute units)
blk = np.arange(n) // 256
orders = {"generated": np.arange(n), "sorted by path": np.argsort(key, kind="stable"),
          "sorted, search first": np.argsort(-key, kind="stable"),
          "within workgroups": np.lexsort((key, blk)), "within workgroups, search first": np.lexsort((-key, blk))}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
plans = {}
for name, order in orders.items():
    m12 = matrices_to_m12_soa(M[order], dev)
    out = {"joints": torch.empty((n, 7), dtype=torch.float64, device=dev), "reachable": torch.empty((n,), dtype=torch.uint8, device=dev),
           "state": torch.empty((n,), dtype=torch.uint8, device=dev), "emergency": torch.empty((n,), dtype=torch.uint8, device=dev)}
    p = c.symbolic_inverse_kinematics_batch("r_arm", m12, out=out, plan_only=True)
    plans[name] = (p, m12, out, order)
best = {k: 1e9 for k in orders}
for rnd in range(4):
    for name, (p, m12, out, order) in plans.items():
        for _ in range(20):
            p["launch"]()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(200):
            p["launch"]()
        e1.record()
        torch.cuda.synchronize()
        best[name] = min(best[name], e0.elapsed_time(e1) / 200 * 1e3)
ref = plans["generated"][2]["joints"].cpu().numpy()
for name, (p, m12, out, order) in plans.items():
    same = np.array_equal(out["joints"].cpu().numpy(), ref[order], equal_nan=True)
    print(f"{name:22s}: {best[name]:6.2f} us per launch (best of 4 x 200), rows identical to the generated order's: {same}")
print(json.dumps({"kernel_us": best, "shares": {"shortcut": float(np.mean(~need & found)), "search": float(need.mean()),
                                                "nothing_found": float(np.mean(need & ~found))}}))
