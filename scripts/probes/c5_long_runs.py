#!/usr/bin/env python3
"""Round 6, review task 1(a): what a LONG continuous run costs per 1000 steps, by block size.

    python scripts/probes/c5_long_runs.py [n_traj] [--steps 1000,2000,...] [--blocks 0,352,...] [--reps 4]

For every run length N and every RSIK_OPT_CONT_BLOCK_STEPS value B (0 = the library's own choice) a run of n_traj x N steps is
issued launch by launch `reps` times (the first is untimed) from the same initial state; printed: ms per run, ms per 1000 steps,
and whether flags / states / joints / the carried theta are bit-identical to the B = 0 run of that length.  The head of a run
(start-up search beside the first prepare, the first theta) and its tail (last joints, last chain) are paid once per run:
the longer the run the closer the figure gets to what the chip-filling phases alone allow.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from reachy2_symbolic_ik_amd import ControlIK, _abi  # noqa: E402

if os.environ.get("C5_LIB"):  # another build of the library (scripts/build_variant.py)
    _abi.use_library(os.path.abspath(os.environ["C5_LIB"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n_traj", nargs="?", type=int, default=4096)
    ap.add_argument("--steps", default="1000,2000,4000,8000,16000")
    ap.add_argument("--blocks", default="0,352")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--graph", action="store_true", help="also time the run captured once and replayed")
    ap.add_argument("--variants", default="0", help="RSIK_OPT_CONT_PHASED_VARIANT values, comma-separated (64 = the joints phase without its turn hints)")
    args = ap.parse_args()
    n = args.n_traj
    lengths = [int(v) for v in args.steps.split(",")]
    blocks = [int(v) for v in args.blocks.split(",")]
    ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
    hs = ctrl._solver
    # one generator call for the longest run: shorter runs are its first N steps (same trajectories)
    traj_all = bench.make_config5_trajectories(n, max(lengths), seed=7, device=0)
    print(f"# {n} trajectories; eager = launch by launch; ms per 1000 steps = ms per run / (N / 1000)", flush=True)
    for N in lengths:
        traj = traj_all[:N].contiguous()
        cont0 = ctrl.new_continuous_state("r_arm", n)
        ref = None
        for B, V in [(b_, v_) for b_ in blocks for v_ in [int(v) for v in args.variants.split(",")]]:
            hs.set_option(_abi.OPT_CONT_BLOCK_STEPS, B)
            hs.set_option(_abi.OPT_CONT_PHASED_VARIANT, V)
            cont = cont0.clone()
            out = None
            best, times = None, []
            for rep in range(args.reps):
                cont.copy_(cont0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = ctrl.run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=out)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3
                if rep > 0:
                    times.append(ms)
            best = min(times)
            med = sorted(times)[len(times) // 2]
            got = (out["reachable"].clone(), out["state"].clone(), out["joints"].clone(), cont[0].clone())
            if ref is None:
                ref, same = got, "reference"
            else:
                bits = torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[3], ref[3])
                dj = float((got[2] - ref[2]).abs().nan_to_num(0.0).max())
                same = ("flags/states/theta identical" if bits else "FLAGS DIFFER") + f", max joint diff {dj:.1e}"
            line = f"N {N:6d}  block {B:5d} variant {V:2d}: best {best:8.3f} ms  median {med:8.3f} ms  -> {best / (N / 1000):.4f} ms per 1000 steps  [{same}]"
            if args.graph:
                try:
                    cont.copy_(cont0)
                    g, _ = ctrl.capture_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=out)
                    gt = []
                    for rep in range(args.reps):
                        cont.copy_(cont0)
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        g.replay()
                        torch.cuda.synchronize()
                        gt.append((time.perf_counter() - t0) * 1e3)
                    line += f"  | replayed best {min(gt[1:]):8.3f} ms -> {min(gt[1:]) / (N / 1000):.4f}"
                    del g
                except Exception as e:  # information only
                    line += f"  | replay failed: {type(e).__name__}: {e}"
            print(line, flush=True)
            del out
        del traj, ref
        torch.cuda.empty_cache()
    hs.set_option(_abi.OPT_CONT_BLOCK_STEPS, 0)
    hs.set_option(_abi.OPT_CONT_PHASED_VARIANT, 0)


if __name__ == "__main__":
    main()
