#!/usr/bin/env python3
"""Counters of the look-back joints phase on config 5 (a -DRSIK_LB_STATS build): how many chunk-trajectories are quiet / walked /
latched, why, and how long the waves poll.   python scripts/build_variant.py lb_stats -DRSIK_LB_STATS;  python scripts/probes/lb_stats.py build/variants/lb_stats.so"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
lib = _abi.load()
buf = (ctypes.c_ulonglong * 16)()
def one():
    cont.copy_(cont0)
    ctrl.run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=out)
one(); one()
lib.rsik_debug_lb_stats(buf, 1)
one()
lib.rsik_debug_lb_stats(buf, 1)
names = ["chunk-trajectories", "mode 0 (raw + turns)", "mode 1 (walked)", "mode 2/3 (latched)", "eventful", "boundary not quiet", "something turned",
         "behind a Pe", "behind a P that is not small", "event inside the chunk", "waves", "raw-word polls (sum)", "look-back polls (sum)"]
for k, nm in enumerate(names):
    print(f"{nm:32s} {buf[k]:12d}   {buf[k] / max(1, buf[0] if k < 10 else buf[10]):.4f}")
