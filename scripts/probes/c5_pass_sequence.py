#!/usr/bin/env python3
"""Round 6: a long free-running sequence of config-5 passes, pass by pass — when does the host's issue stall, and what does the device
do meanwhile?   python scripts/probes/c5_pass_sequence.py [passes] [resident 0|1]

Per pass: the host's issue time (the call's return) and the device time between the events recorded behind consecutive passes.
Printed: the summary and every pass whose host or device time is more than twice the median."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from reachy2_symbolic_ik_amd import ControlIK  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
resident = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
n, n_steps = 4096, 1000
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
traj = bench.make_config5_trajectories(n, n_steps, seed=20250204, device=0)
cont0 = ctrl.new_continuous_state("r_arm", n)
cont = cont0.clone()
out = None
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
for e in ev:
    e.record()
torch.cuda.synchronize()
host = []
ev[0].record()
for k in range(N):
    t0 = time.perf_counter()
    cont.copy_(cont0)
    out = ctrl.run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=out, goals_resident=resident)
    ev[k + 1].record()
    host.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
dev = [ev[k].elapsed_time(ev[k + 1]) for k in range(N)]
import statistics

mh, md = statistics.median(host), statistics.median(dev)
print(f"{N} passes, resident={resident}: host issue median {mh:.3f} ms, device median {md:.4f} ms per pass; totals host {sum(host):.1f} ms device {sum(dev):.1f} ms")
for a in range(0, N, 20):
    print(f"  passes {a:4d}-{min(a + 19, N - 1):4d}: device mean {statistics.mean(dev[a:a + 20]):.4f} ms  host mean {statistics.mean(host[a:a + 20]):.3f} ms  host max {max(host[a:a + 20]):.2f}")
odd = [(k, host[k], dev[k]) for k in range(N) if host[k] > 3 * mh or dev[k] > 2 * md]
print("outliers (pass, host ms, device ms):", [(k, round(h, 2), round(d, 3)) for k, h, d in odd][:40])
