#!/usr/bin/env python3
"""Timeline of the continuous pipeline's phase kernels WITHOUT a profiler: a -DRSIK_PIPE_TIMING build stamps every phase
kernel's first start and last end with the 100 MHz counter (see rsik_kernel_pipeline.hpp).

    python scripts/build_variant.py pipe_timing -DRSIK_PIPE_TIMING
    RSIK_PIPE_TIMING_PRINT=1 python scripts/probes/c5_untraced_timeline.py build/variants/pipe_timing.so [steps per block] [n_steps]
(stamps exist for the first 64 blocks of a run; C5_GRAPH=0 skips the replayed form; C5_RESIDENT=1: the passes overlap,
RSIK_OPT_CONT_GOALS_RESIDENT; C5_OWN_STREAM=1: on a torch stream of its own instead of the NULL stream — the stamp area is cleared on the caller's stream at a run's start, so the stamps of prepare kernels
that ran before that are lost or partial)
"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reachy2_symbolic_ik_amd import _abi
_abi.use_library(os.path.abspath(sys.argv[1]))
import bench
from reachy2_symbolic_ik_amd import ControlIK
blk = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n, n_steps = 4096, (int(sys.argv[3]) if len(sys.argv) > 3 else 1000)
traj = bench.make_config5_trajectories(n, n_steps, seed=20250204, device=0)
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
ctrl._solver.set_option(_abi.OPT_CONT_BLOCK_STEPS, blk)
if os.environ.get("C5_OWN_STREAM"):
    torch.cuda.set_stream(torch.cuda.Stream())
cont0 = ctrl.new_continuous_state("r_arm", n)
out = {"joints": torch.empty((n_steps, n, 7), dtype=torch.float64, device="cuda"),
       "reachable": torch.empty((n_steps, n), dtype=torch.uint8, device="cuda"),
       "state": torch.empty((n_steps, n), dtype=torch.uint8, device="cuda")}
cont = cont0.clone()
os.environ.pop("RSIK_PIPE_TIMING_PRINT", None)
def one():
    cont.copy_(cont0)
    ctrl.run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=out,
                                     goals_resident=bool(int(os.environ.get("C5_RESIDENT", "0"))))
for _ in range(6): one()
torch.cuda.synchronize()
t0 = time.perf_counter()
host = 0.0
for _ in range(10):
    h0 = time.perf_counter(); one(); host += time.perf_counter() - h0
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per pass (eager, stamps on); the host's calls return after {host / 10 * 1e3:.3f} ms", file=sys.stderr)
os.environ["RSIK_PIPE_TIMING_PRINT"] = "1"
one()   # prints the stamps of the pass before it
torch.cuda.synchronize()
# the same for a pass replayed from a hipGraph: replay, then one eager call that prints the stamps the replay left
if os.environ.get("C5_GRAPH", "1") == "1":
    os.environ.pop("RSIK_PIPE_TIMING_PRINT", None)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        one(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            one()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per pass (graph replay, stamps on)", file=sys.stderr)
    print("---- graph replay ----", file=sys.stderr)
    os.environ["RSIK_PIPE_TIMING_PRINT"] = "1"
    one()
    torch.cuda.synchronize()
