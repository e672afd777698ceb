#!/usr/bin/env python3
"""Config 5 (4096 trajectories x 1000 control steps per pass) with several builds of the library, one process each
(scripts/c5_libs.py libA.so libB.so ...): ms per pass and steps/s, two rounds; then every library's results against the
first one's: flags, state codes and the carried state bit for bit, the largest difference between joints."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(%r))
from reachy2_symbolic_ik_amd import _abi
_abi.use_library(os.path.abspath(sys.argv[1]))
import bench
from reachy2_symbolic_ik_amd import ControlIK
n, n_steps = 4096, 1000
traj = bench.make_config5_trajectories(n, n_steps, seed=20250204, device=0)
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
cont0 = ctrl.new_continuous_state("r_arm", n)
out = {"joints": torch.empty((n_steps, n, 7), dtype=torch.float64, device="cuda"),
       "reachable": torch.empty((n_steps, n), dtype=torch.uint8, device="cuda"),
       "state": torch.empty((n_steps, n), dtype=torch.uint8, device="cuda")}
cont = cont0.clone()
def one():
    cont.copy_(cont0)
    ctrl.run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=out)
for _ in range(5): one()
best = 1e9
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): one()
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
print(f"{sys.argv[1]}: {best:.3f} ms per pass, {n*n_steps/best/1e6:.2f} G steps/s, checksum {float(out['joints'].sum()):.12e}", flush=True)
if len(sys.argv) > 2:
    torch.save({"joints": out["joints"].cpu(), "reachable": out["reachable"].cpu(), "state": out["state"].cpu(), "cont": cont.cpu()}, sys.argv[2])
''' % HERE
libs = sys.argv[1:]
for round_ in range(2):
    for k, lib in enumerate(libs):
        subprocess.run([sys.executable, "-c", CHILD, lib] + ([f"/tmp/c5_libs_{k}.pt"] if round_ == 1 else []), check=False)
import torch  # noqa: E402

ref = torch.load("/tmp/c5_libs_0.pt")
for k, lib in enumerate(libs[1:], start=1):
    got = torch.load(f"/tmp/c5_libs_{k}.pt")
    bits = {name: bool(torch.equal(ref[name].view(torch.uint8), got[name].view(torch.uint8))) for name in ("reachable", "state", "cont")}
    dj = (ref["joints"] - got["joints"]).abs()
    nan_same = bool(torch.equal(ref["joints"].isnan(), got["joints"].isnan()))
    print(f"{lib} vs {libs[0]}: bit-identical {bits}, carried theta identical {bool(torch.equal(ref['cont'][0], got['cont'][0]))}, "
          f"max |joints difference| {float(dj.nan_to_num().max()):.3e} rad (NaN pattern identical: {nan_same}), "
          f"joints bit-identical in {float((dj == 0).double().mean()) * 100:.2f} % of the entries")
