#!/usr/bin/env python3
"""bench.py — IK solves/s of the MI355X-native analytic solve path.

    python bench.py --gpus N --steps K --warmup W [--config 2|3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one pass of the hot path over one resident batch:
  config 2 (default, the configuration BASELINE.json's metric is quoted on): r_arm
    SymbolicIK.is_reachable + theta_to_joints_func(theta = interval[0]) on 1 048 576 random REACHABLE poses per GPU
    (SoA float64 in HBM), outputs joints [n,7], interval [n,2], reachable, state  -> 122 algorithmic B/pose.
  config 3: ControlIK discrete mode, 64-point elbow sweep, 262 144 wrist-reachable goal matrices per GPU -> 154 B/pose.
With N > 1 every rank solves its own shard (weak scaling) and the step ends with the RCCL all-gather of the joint
array (+ flags) that the north star names; the kernel-only rate is reported next to it.

Prints ONE JSON line on rank 0 (see the repo prompt's bench contract) carrying `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
BYTES_PER_POSE = {2: 48 + 56 + 16 + 1 + 1, 3: 96 + 56 + 1 + 1}
URDF = "config_files/reachy2_ik_minimal.urdf"
SHOULDER_R = np.array([0.0, -0.2, 0.0])


def _quiet(fn, *a, **k):
    import contextlib
    import io

    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def make_config2_poses(n, seed=20250204, device=0):
    """pos = s_r + U(-0.7,0.7)^3, eul = U(-pi,pi)^3, keep the first n whose is_reachable state
    is "reachable".  The filter is the product's own HIP kernel (theta policy "none")."""
    import torch

    from reachy2_symbolic_ik_amd import SymbolicIK

    ik = _quiet(SymbolicIK, "r_arm", device=device)
    rng = np.random.default_rng(seed)
    P, E, have = [], [], 0
    chunk = 1 << 22
    while have < n:
        pos = SHOULDER_R + rng.uniform(-0.7, 0.7, size=(chunk, 3))
        eul = rng.uniform(-np.pi, np.pi, size=(chunk, 3))
        soa = torch.as_tensor(np.ascontiguousarray(np.concatenate([pos.T, eul.T], axis=0))).to(ik.solver.device)
        ok = ik.is_reachable_batch(soa)["reachable"].cpu().numpy().astype(bool)
        P.append(pos[ok])
        E.append(eul[ok])
        have += int(ok.sum())
    return np.concatenate(P)[:n].copy(), np.concatenate(E)[:n].copy()


def make_config3_matrices(n, seed=20250204, device=0):
    """goal matrices from the same generator, kept when the (non-DVT ControlIK) solver's
    is_reachable state is "reachable" — NOT filtered by the elbow test, so the sweep really runs."""
    import torch

    from reachy2_symbolic_ik_amd import SymbolicIK
    from reachy2_symbolic_ik_amd.constants import euler_xyz_extrinsic

    ik = _quiet(SymbolicIK, "r_arm", singularity_offset=-1.01, device=device)
    rng = np.random.default_rng(seed + 3)
    P, E, have = [], [], 0
    chunk = 1 << 21
    while have < n:
        pos = SHOULDER_R + rng.uniform(-0.7, 0.7, size=(chunk, 3))
        eul = rng.uniform(-np.pi, np.pi, size=(chunk, 3))
        soa = torch.as_tensor(np.ascontiguousarray(np.concatenate([pos.T, eul.T], axis=0))).to(ik.solver.device)
        ok = ik.is_reachable_batch(soa)["reachable"].cpu().numpy().astype(bool)
        P.append(pos[ok])
        E.append(eul[ok])
        have += int(ok.sum())
    pos, eul = np.concatenate(P)[:n], np.concatenate(E)[:n]
    # vectorised Rz(c) Ry(b) Rx(a)
    ca, sa, cb, sb, cc, sc = np.cos(eul[:, 0]), np.sin(eul[:, 0]), np.cos(eul[:, 1]), np.sin(eul[:, 1]), np.cos(eul[:, 2]), np.sin(eul[:, 2])
    M = np.zeros((n, 4, 4))
    M[:, 0, 0] = cc * cb; M[:, 0, 1] = cc * sb * sa - sc * ca; M[:, 0, 2] = cc * sb * ca + sc * sa
    M[:, 1, 0] = sc * cb; M[:, 1, 1] = sc * sb * sa + cc * ca; M[:, 1, 2] = sc * sb * ca - cc * sa
    M[:, 2, 0] = -sb; M[:, 2, 1] = cb * sa; M[:, 2, 2] = cb * ca
    M[:, :3, 3] = pos
    M[:, 3, 3] = 1.0
    assert np.allclose(M[0, :3, :3], euler_xyz_extrinsic(eul[0]))
    return M


def cpu_baseline(config, inputs, seconds):
    """Times the CPU checker (oracle/, a C restatement of the reference path = kind "port") on the host cores,
    on a bounded sample of the SAME workload.  Reported baseline, not the target."""
    from oracle import oracle as orc

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    avail = max(1, min(avail, orc.lib().orc_max_threads()))
    if config == 2:
        pos, eul = inputs
        m = min(len(pos), 1 << 18)
        pos, eul = np.ascontiguousarray(pos[:m]), np.ascontiguousarray(eul[:m])
        ar, al = orc.Arm("r_arm", 0.03), orc.Arm("l_arm", 0.03)
        run = lambda nt: orc.solve_batch(ar, al, pos, eul, nthreads=nt)  # noqa: E731
    else:
        M = np.ascontiguousarray(inputs[: min(len(inputs), 1 << 17)])
        m = len(M)
        ar, al = orc.Arm("r_arm", -1.01), orc.Arm("l_arm", -1.01)
        run = lambda nt: orc.control_discrete_batch(ar, al, M, nb_search_points=64, nthreads=nt)  # noqa: E731

    def rate(nt, budget):
        run(nt)
        t0 = time.perf_counter()
        passes = 0
        while True:
            run(nt)
            passes += 1
            el = time.perf_counter() - t0
            if el >= budget or passes >= 2000:
                return passes * m / el, passes, el

    # the host may expose more logical CPUs than the container's CPU quota: probe a few thread counts briefly,
    # then spend the remaining budget on the best one
    cands = sorted({c for c in (1, 8, 16, 32, 64, 128, avail) if c <= avail})
    probe = {c: rate(c, 0.7)[0] for c in cands}
    best = max(probe, key=probe.get)
    value, passes, el = rate(best, max(1.0, seconds - 0.7 * len(cands)))
    return {
        "value": value,
        "unit": "solves/s",
        "cores": best,
        "kind": "port",
        "single_thread": probe[1],
        "sample": f"first {m} poses of the workload x {passes} passes ({el:.1f} s) with OpenMP {best} threads "
                  f"(best of {cands}; 1 thread: {probe[1]:.0f} solves/s); C restatement built gcc -O2 -ffp-contract=off",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3])
    ap.add_argument("--poses", type=int, default=0, help="poses per GPU (default: the BASELINE size)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    from reachy2_symbolic_ik_amd import ControlIK, SymbolicIK

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group(backend="nccl", device_id=dev)
    n_gpus = world
    cfg = args.config
    n = args.poses or ((1 << 20) if cfg == 2 else (1 << 18))

    # ---- synthetic inputs, resident in HBM before the timed region
    if cfg == 2:
        pos, eul = make_config2_poses(n, seed=20250204 + rank, device=local_rank)
        inputs = (pos, eul)
        ik = _quiet(SymbolicIK, "r_arm", device=local_rank)
        soa = torch.as_tensor(np.ascontiguousarray(np.concatenate([pos.T, eul.T], axis=0))).to(dev)
        assert soa.is_contiguous()
        out = {
            "joints": torch.empty((n, 7), dtype=torch.float64, device=dev),
            "interval": torch.empty((n, 2), dtype=torch.float64, device=dev),
            "reachable": torch.empty((n,), dtype=torch.uint8, device=dev),
            "state": torch.empty((n,), dtype=torch.uint8, device=dev),
        }
        plan = ik.solve_batch(soa, want_elbow=False, out=out, plan_only=True)
        step_kernel = plan["launch"]  # one rsik_solve call with pre-bound arguments
        workload = f"config2: r_arm is_reachable + theta_to_joints_func(interval[0]), {n} random reachable poses per GPU"
        kernel_name = "solve_kernel"
    else:
        M = make_config3_matrices(n, seed=20250204 + rank, device=local_rank)
        inputs = M
        ctrl = _quiet(ControlIK, urdf_path=URDF, device=local_rank)
        ctrl.nb_search_points = 64
        from reachy2_symbolic_ik_amd.control_ik import matrices_to_m12_soa

        m12 = matrices_to_m12_soa(M, dev)
        out = {
            "joints": torch.empty((n, 7), dtype=torch.float64, device=dev),
            "reachable": torch.empty((n,), dtype=torch.uint8, device=dev),
            "state": torch.empty((n,), dtype=torch.uint8, device=dev),
            "emergency": torch.empty((n,), dtype=torch.uint8, device=dev),
        }
        plan = ctrl.symbolic_inverse_kinematics_batch("r_arm", m12, out=out, plan_only=True)
        step_kernel = plan["launch"]  # one rsik_control_discrete call with pre-bound arguments
        workload = f"config3: r_arm ControlIK discrete, 64-point theta sweep, {n} wrist-reachable goal matrices per GPU"
        kernel_name = "control_discrete_kernel"

    gathered = None
    if world > 1:
        gathered = {
            "joints": torch.empty((world * n, 7), dtype=torch.float64, device=dev),
            "reachable": torch.empty((world * n,), dtype=torch.uint8, device=dev),
        }

    def step():
        step_kernel()
        if world > 1:  # the final joint array (+ flags) is all-gathered over xGMI
            dist.all_gather_into_tensor(gathered["joints"], out["joints"])
            dist.all_gather_into_tensor(gathered["reachable"], out["reachable"])

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # ---- timed region: exactly K steps; per-launch kernel time from events on the launch stream
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        step_kernel()
        ev[k][1].record()
        if world > 1:
            dist.all_gather_into_tensor(gathered["joints"], out["joints"])
            dist.all_gather_into_tensor(gathered["reachable"], out["reachable"])
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    if world > 1:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    # sanity: the timed outputs are real results (flags all "reachable" for config 2)
    n_ok = int(out["reachable"].sum().item())
    if cfg == 2:
        assert n_ok == n, f"{n - n_ok} poses of the reachable workload came back unreachable"
        assert bool(torch.isfinite(out["joints"]).all()), "non-finite joints"

    if rank == 0:
        total = n * n_gpus * args.steps
        value = total / elapsed
        bpp = BYTES_PER_POSE[cfg]
        achieved = bpp * n / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "IK solves/sec (7-DoF r_arm, batched poses)",
            "value": value,
            "unit": "solves/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "poses_per_gpu": n, "theta_policy": "interval[0]" if cfg == 2 else "discrete sweep nb=64",
                       "collective": "none" if world == 1 else "RCCL all-gather of joints [n,7] f64 + reachable u8 per step"},
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_name,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "algorithmic_bytes_per_pose": bpp,
                "kernel_ms": kernel_ms,
                "kernel_only_solves_per_s_per_gpu": n / (kernel_ms * 1e-3),
                "note": "fp64 VALU-bound path (see DESIGN.md): HBM fraction is reported as the contract asks, VALU issue is the binding limit",
            },
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, inputs, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
