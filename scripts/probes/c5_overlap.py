#!/usr/bin/env python3
"""Round 6, review task 1(c): consecutive continuous runs that overlap (RSIK_OPT_CONT_GOALS_RESIDENT).

    python scripts/probes/c5_overlap.py [n_traj] [n_steps] [--k 20] [--rounds 3] [--blocks 0]

The bench's config-5 protocol (every pass resets the trajectory state and re-initialises, same goals, same output buffers), K passes
back to back, timed with HIP events on the launch stream, in four forms, `rounds` times alternately:
  serial     launch by launch, every run's streams meet at its start and end (the default)
  resident   launch by launch with the promise set: the prepare phase of pass k + 1 beside the tail of pass k
  resident2  the same with two sets of output buffers taking turns (no row of pass k + 1 waits for pass k's chain kernel)
  graph      one captured pass replayed K times
and `isolated`: serial passes with a device synchronisation behind each.  The last pass's outputs and the trajectory state are
compared with the serial form's bit for bit.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from reachy2_symbolic_ik_amd import ControlIK, _abi  # noqa: E402

if os.environ.get("C5_LIB"):  # another build of the library (scripts/build_variant.py)
    _abi.use_library(os.path.abspath(os.environ["C5_LIB"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n_traj", nargs="?", type=int, default=4096)
    ap.add_argument("n_steps", nargs="?", type=int, default=1000)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=0)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--variant", type=int, default=0, help="RSIK_OPT_CONT_PHASED_VARIANT")
    ap.add_argument("--own-stream", action="store_true", help="everything on a torch stream of its own instead of the NULL stream")
    args = ap.parse_args()
    n, n_steps, K = args.n_traj, args.n_steps, args.k
    dev = torch.device("cuda", 0)
    ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
    hs = ctrl._solver
    hs.set_option(_abi.OPT_CONT_BLOCK_STEPS, args.blocks)
    hs.set_option(_abi.OPT_CONT_PHASED_VARIANT, args.variant)
    traj = bench.make_config5_trajectories(n, n_steps, seed=20250204, device=0)
    cont0 = ctrl.new_continuous_state("r_arm", n)
    cont = cont0.clone()

    def new_out():
        return {"joints": torch.empty((n_steps, n, 7), dtype=torch.float64, device=dev),
                "reachable": torch.empty((n_steps, n), dtype=torch.uint8, device=dev),
                "state": torch.empty((n_steps, n), dtype=torch.uint8, device=dev)}

    outs = [new_out(), new_out()]
    forms_seen = {}

    def one(resident, out):
        cont.copy_(cont0)
        r = ctrl.run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=out, goals_resident=resident)
        forms_seen[r.run_form_name] = forms_seen.get(r.run_form_name, 0) + 1

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    host_ms = {}

    def timed(fn, k, name=None):
        import time

        torch.cuda.synchronize()
        e0.record()
        t0 = time.perf_counter()
        for i in range(k):
            fn(i)
        host = (time.perf_counter() - t0) * 1e3 / k  # what the host needs to ISSUE a pass (its calls return: nothing waits for the device)
        e1.record()
        torch.cuda.synchronize()
        if name:
            host_ms.setdefault(name, []).append(host)
        return e0.elapsed_time(e1) / k

    def snapshot(out):
        return {k: v.clone() for k, v in out.items()}, cont.clone()

    def same(a, b):
        flags = all(torch.equal(a[0][k], b[0][k]) for k in ("reachable", "state")) and torch.equal(a[1], b[1])
        dj = float((a[0]["joints"] - b[0]["joints"]).abs().nan_to_num(0.0).max())
        nan_same = torch.equal(torch.isnan(a[0]["joints"]), torch.isnan(b[0]["joints"]))
        return ("bit-identical flags / states / trajectory state" if flags else "FLAGS OR STATE DIFFER") + f", max joint diff {dj:.1e}" + ("" if nan_same else ", NaN PATTERN DIFFERS")

    graph = None
    if not args.no_graph:
        cont.copy_(cont0)
        # (a context of its own: one a hipGraph points into never overlaps its launch-by-launch runs — the library cannot see replays)
        ctrl_g = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
        ctrl_g._solver.set_option(_abi.OPT_CONT_BLOCK_STEPS, args.blocks)
        graph, _ = ctrl_g.capture_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=outs[0])

    def graph_pass(i):
        cont.copy_(cont0)
        graph.replay()

    forms = {
        "serial": lambda i: one(False, outs[0]),
        "resident": lambda i: one(True, outs[0]),
        "resident2": lambda i: one(True, outs[i & 1]),
    }
    if graph is not None:
        forms["graph"] = graph_pass
    for name, fn in forms.items():  # warm-up
        timed(fn, 5)
    ref = None
    results = {k: [] for k in forms}
    iso = []
    for rnd in range(args.rounds):
        for name, fn in forms.items():
            ms = timed(fn, K, name)
            results[name].append(ms)
            snap = snapshot(outs[(K - 1) & 1] if name == "resident2" else outs[0])
            if name == "serial" and ref is None:
                ref = snap
            verdict = "reference" if (name == "serial" and rnd == 0) else same(snap, ref)
            print(f"round {rnd}  {name:10s} {ms:7.4f} ms per pass  {n * n_steps / ms / 1e6:6.2f} G steps/s   [{verdict}]", flush=True)
        t = 0.0
        for i in range(K):
            torch.cuda.synchronize()
            e0.record()
            one(False, outs[0])
            e1.record()
            torch.cuda.synchronize()
            t += e0.elapsed_time(e1)
        iso.append(t / K)
        print(f"round {rnd}  isolated   {t / K:7.4f} ms per pass (a device synchronisation behind every pass)", flush=True)
    print("host issue time per pass (ms, best):", {k: round(min(v), 4) for k, v in host_ms.items()})
    print("run forms seen:", forms_seen)
    print("best per form:", {k: round(min(v), 4) for k, v in results.items()}, "isolated", round(min(iso), 4))


if __name__ == "__main__":
    if "--own-stream" in sys.argv:
        with torch.cuda.stream(torch.cuda.Stream()):
            main()
    else:
        main()
