#!/usr/bin/env python3
"""Config 5, eager and replayed from a hipGraph, for several builds of the library (one process each), interleaved:
scripts/probes/c5_graph_libs.py libA.so libB.so ..."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(%r)))
from reachy2_symbolic_ik_amd import _abi
_abi.use_library(os.path.abspath(sys.argv[1]))
import bench
from reachy2_symbolic_ik_amd import ControlIK
n, n_steps = 4096, 1000
traj = bench.make_config5_trajectories(n, n_steps, seed=20250204, device=0)
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
ctrl._solver.set_option(_abi.OPT_CONT_BLOCK_STEPS, int(os.environ.get("C5_BLOCK", "0")))
cont0 = ctrl.new_continuous_state("r_arm", n)
out = {"joints": torch.empty((n_steps, n, 7), dtype=torch.float64, device="cuda"),
       "reachable": torch.empty((n_steps, n), dtype=torch.uint8, device="cuda"),
       "state": torch.empty((n_steps, n), dtype=torch.uint8, device="cuda")}
cont = cont0.clone()
def one():
    cont.copy_(cont0)
    ctrl.run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=out)
def timed(f, reps=20):
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best
for _ in range(5): one()
torch.cuda.synchronize()
e = timed(one)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    one(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        one()
g.replay(); torch.cuda.synchronize()
print(f"{sys.argv[1]} block {os.environ.get('C5_BLOCK', '0')}: eager {e:.3f} ms, graph {timed(g.replay):.3f} ms per pass")
''' % HERE
for round_ in range(3):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", CHILD, lib], check=False)
