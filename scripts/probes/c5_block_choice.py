#!/usr/bin/env python3
"""Config 5 (4096 x 1000): eager passes and replayed graphs by block size (RSIK_OPT_CONT_BLOCK_STEPS), interleaved rounds — which cut of a
run into blocks the defaults should make.  usage: c5_block_choice.py "320,336,352,..." [rounds]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from reachy2_symbolic_ik_amd import ControlIK, _abi as A
blocks = [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "0,336,352,368,384,400,512").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n, T = 4096, 1000
traj = bench.make_config5_trajectories(n, T, seed=20250204, device=0)
c = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
c._solver.control_continuous_reserve(n, T)
cont0 = c.new_continuous_state("r_arm", n)
out = {"joints": torch.empty((T, n, 7), dtype=torch.float64, device="cuda"), "reachable": torch.empty((T, n), dtype=torch.uint8, device="cuda"),
       "state": torch.empty((T, n), dtype=torch.uint8, device="cuda")}
cont = cont0.clone()
def one():
    cont.copy_(cont0)
    c.run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=out)
def timed(f, reps=20):
    for _ in range(4): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
graphs = {}
s = torch.cuda.Stream()
for blk in blocks:
    c._solver.set_option(A.OPT_CONT_BLOCK_STEPS, blk)
    with torch.cuda.stream(s):
        for _ in range(2): one()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            one()
    graphs[blk] = g
res = {}
for r in range(rounds):
    for blk in blocks:
        c._solver.set_option(A.OPT_CONT_BLOCK_STEPS, blk)
        e = timed(one); gms = timed(graphs[blk].replay)
        res.setdefault(blk, []).append((e, gms))
        print(f"round {r} block {blk:4d}: eager {e:.4f}  replayed {gms:.4f} ms per pass", flush=True)
for blk, v in res.items():
    print(f"== block {blk:4d}: eager min {min(x[0] for x in v):.4f} median {sorted(x[0] for x in v)[len(v)//2]:.4f} | replayed min {min(x[1] for x in v):.4f} median {sorted(x[1] for x in v)[len(v)//2]:.4f}")
