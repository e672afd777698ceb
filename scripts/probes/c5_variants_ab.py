#!/usr/bin/env python3
"""Config 5 (4096 x 1000), eager passes under several RSIK_OPT_CONT_PHASED_VARIANT values / block sizes, interleaved rounds, results
compared bit for bit with the first: usage c5_variants_ab.py "variant:block,variant:block,..." [rounds]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from reachy2_symbolic_ik_amd import ControlIK, _abi as A
cases = [tuple(int(x) for x in c.split(":")) for c in (sys.argv[1] if len(sys.argv) > 1 else "0:0,4:0,12:0").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n, T = 4096, 1000
traj = bench.make_config5_trajectories(n, T, seed=20250204, device=0)
c = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
cont0 = c.new_continuous_state("r_arm", n)
out = {"joints": torch.empty((T, n, 7), dtype=torch.float64, device="cuda"), "reachable": torch.empty((T, n), dtype=torch.uint8, device="cuda"),
       "state": torch.empty((T, n), dtype=torch.uint8, device="cuda")}
cont = cont0.clone()
def one():
    cont.copy_(cont0)
    c.run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=out)
ref = None
best = {}
for r in range(rounds):
    for v, blk in cases:
        c._solver.set_option(A.OPT_CONT_PHASED_VARIANT, v)
        c._solver.set_option(A.OPT_CONT_BLOCK_STEPS, blk)
        for _ in range(5): one()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): one()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
        got = {k: x.clone() for k, x in out.items()}; got["cont"] = cont.clone()
        if ref is None: ref = got
        same = all(torch.equal(ref[k].view(torch.uint8), got[k].view(torch.uint8)) for k in ref)
        best.setdefault((v, blk), []).append(ms)
        print(f"round {r} variant {v:2d} block {blk:4d}: {ms:.4f} ms per pass  [{'bit-identical' if same else 'DIFFERENT'}]", flush=True)
for k, v in best.items():
    print(f"== variant {k[0]:2d} block {k[1]:4d}: min {min(v):.4f}  median {sorted(v)[len(v)//2]:.4f} ms")
