#!/usr/bin/env python3
"""PCIe-inclusive rate of the solve path for a batch that lives in host memory:
one upload + one launch + one download against SymbolicIK.solve_batch_host's chunked 3-stream pipeline.

    python scripts/host_pipeline.py [n_poses] [chunk]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from reachy2_symbolic_ik_amd import SymbolicIK  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
ik = bench._quiet(SymbolicIK, "r_arm")
P, E = bench.make_config2_poses(min(n, 1 << 20))
reps = (n + len(P) - 1) // len(P)
soa = np.ascontiguousarray(np.concatenate([np.tile(P, (reps, 1))[:n].T, np.tile(E, (reps, 1))[:n].T], axis=0))
host = torch.as_tensor(soa).pin_memory()
out = {"joints": torch.empty((n, 7), dtype=torch.float64).pin_memory(), "interval": torch.empty((n, 2), dtype=torch.float64).pin_memory(),
       "reachable": torch.empty((n,), dtype=torch.uint8).pin_memory(), "state": torch.empty((n,), dtype=torch.uint8).pin_memory()}


def naive():
    d = host.cuda(non_blocking=True)
    r = ik.solve_batch(d, want_elbow=False)
    for k in out:
        out[k].copy_(r[k], non_blocking=True)
    torch.cuda.synchronize()


def piped():
    ik.solve_batch_host(host, chunk=chunk, out=out)


def whole():
    ik.solve_batch_host(host, out=out)


def timed(fn, reps=5):
    fn()
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


t_naive = timed(naive)
ref = {k: v.clone() for k, v in out.items()}
t_pipe = timed(piped)
same = all(torch.equal(ref[k], out[k]) if k != "joints" and k != "interval" else torch.allclose(ref[k], out[k], rtol=0, atol=0, equal_nan=True) for k in out)
gb = n * 122 / 1e9
print(f"n = {n} poses in pinned host memory ({n * 48 / 1e6:.0f} MB up, {n * 74 / 1e6:.0f} MB down), chunk = {chunk}")
print(f"upload, solve, download in sequence : {t_naive * 1e3:8.2f} ms  {n / t_naive / 1e9:6.3f} G solves/s  ({gb / t_naive:5.1f} GB/s over PCIe)")
print(f"chunked 3-stream pipeline           : {t_pipe * 1e3:8.2f} ms  {n / t_pipe / 1e9:6.3f} G solves/s  ({gb / t_pipe:5.1f} GB/s over PCIe)")
t_whole = timed(whole)
print(f"solve_batch_host (one chunk)        : {t_whole * 1e3:8.2f} ms  {n / t_whole / 1e9:6.3f} G solves/s  ({gb / t_whole:5.1f} GB/s over PCIe)")
print("results identical:", same)
