"""Timing split of the fused solve on the config-2 workload: reach only (theta policy 'none') vs reach + joints."""
import contextlib, io, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from bench import make_config2_poses
from reachy2_symbolic_ik_amd import SymbolicIK
with contextlib.redirect_stdout(io.StringIO()):
    ik = SymbolicIK("r_arm")
pos, eul = make_config2_poses(1 << 20)
soa = torch.as_tensor(np.ascontiguousarray(np.concatenate([pos.T, eul.T], axis=0))).cuda()
def timed(fn, k=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1000
p_none = ik.solve_batch(soa, theta="none", plan_only=True)["launch"]
p_i0 = ik.solve_batch(soa, want_elbow=False, plan_only=True)["launch"]
p_i0e = ik.solve_batch(soa, want_elbow=True, plan_only=True)["launch"]
frac = torch.rand(1 << 20, dtype=torch.float64, device="cuda")
p_fr = ik.solve_batch(soa, theta=("fraction", frac), want_elbow=False, plan_only=True)["launch"]
print("reach only        %.1f us" % timed(p_none))
print("reach+joints i0   %.1f us" % timed(p_i0))
print("  + elbow output  %.1f us" % timed(p_i0e))
print("reach+joints frac %.1f us" % timed(p_fr))
