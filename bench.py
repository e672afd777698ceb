#!/usr/bin/env python3
"""bench.py — IK solves/s of the MI355X-native analytic solve path.

    python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5]

`--gpus N` with N > 1 starts the N ranks itself (one child process per GPU, RCCL rendezvous on 127.0.0.1) unless it is
already running under torchrun (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`), in which
case RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment.  The parent never touches the GPU.

One "step" = one pass of the hot path over one resident batch:
  config 2 (default at N = 1, the configuration BASELINE.json's metric is quoted on): r_arm SymbolicIK.is_reachable +
    theta_to_joints_func(theta = interval[0]) on 1 048 576 random REACHABLE poses per GPU (SoA float64 in HBM), outputs
    joints [n,7], interval [n,2], reachable, state -> 122 algorithmic B/pose.
  config 3: ControlIK discrete mode, 64-point elbow sweep, 262 144 wrist-reachable goal matrices per GPU -> 154 B/pose.
  config 4 (default at N > 1, BASELINE.json's multi-GPU configuration): r_arm + l_arm mixed batch (per-pose arm byte),
    1 048 576 poses per GPU = 8 M poses at N = 8; each GPU solves its shard K times and the joint array (+ state byte) is
    all-gathered over xGMI with RCCL ONCE, after the last step, inside the timed region (`--gather final`, the default since
    round 5: the north star's job shape); `--gather step` puts a stripe-pipelined all-gather inside EVERY step (`--chunks`; the
    default of rounds 1-4), `--gather none` none.  Both job shapes are timed in every N > 1 run (`multi_gpu.gather_final` /
    `gather_step`, `value_definition` says which one `value` is), beside kernel-only and gather-only figures.
    `--gpus 1 --grouped` takes this code path on a one-rank RCCL group (all a one-GPU box can run of it).
  config 5: ControlIK continuous mode, 4096 trajectories x 1000 control steps per pass; consecutive passes overlap
    (RSIK_OPT_CONT_GOALS_RESIDENT; `steady_state.launch_forms_ms` holds the other launch forms).

Prints ONE JSON line on rank 0 (the repo prompt's bench contract) carrying `roofline` (traffic measured in the run; `extras.hbm_copy`
confirms the 8 TB/s it is priced against) and `cpu_baseline`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

# device-resident kernel-argument buffers (read by the HIP runtime at initialisation; see reachy2_symbolic_ik_amd/__init__.py)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
XGMI_LINK_GBS, XGMI_LINKS = 153.0, 7  # per-direction per-link rate and links per GPU (point-to-point mesh)
NOMINAL_VALU_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 4  # 256 CUs x 4 SIMDs, one wave64 vector instruction per 4 cycles at 2.4 GHz
# SURVEY 8(d).  Config 5: 96 in + 58 out per control step for rsik_control_continuous_run (the trajectory state crosses HBM once
# per pass: the "persistent loop" figure); a caller that launches rsik_control_continuous_step per step also moves the 66-byte
# state in and out every step (286 B) — both fractions are in the line, the headline one is on 154.
BYTES_PER_POSE = {2: 48 + 56 + 16 + 1 + 1, 3: 96 + 56 + 1 + 1, 4: 49 + 56 + 16 + 1 + 1, 5: 96 + 58}
BYTES_PER_STEP_STATE_ROUND_TRIP = 96 + 58 + 2 * 66
GATHER_BYTES_PER_POSE = 56 + 1  # joints [7] f64 + state u8 (reachable == (state == 0) for rsik_solve)
URDF = "config_files/reachy2_ik_minimal.urdf"
SHOULDER_R = np.array([0.0, -0.2, 0.0])
PROFILE_COUNTERS = os.path.join(ROOT, "profiles", "r06", "counters.json")


def _quiet(fn, *a, **k):
    import contextlib
    import io

    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


# One-GPU figures the driver recorded for the configs the N > 1 line runs (BENCH_r03.json, other_configs, W = 5 / K = 20): what an N-GPU
# run's per-GPU kernel rate can be compared with when no one-GPU run of the same job is at hand.  Keyed by (config, rows per GPU).
RECORDED_SINGLE_GPU = {
    (4, 1 << 20): {"solves_per_s": 3.1936e10, "kernel_ms": 0.031708, "source": "BENCH_r03.json other_configs.4 (driver, round 3, one MI355X)"},
}

# the reference itself (single-threaded Python + NumPy), timed in the survey/build container.  Static: the
# reference does not travel to the GPU box.  Carried in every line's cpu_baseline so the true-reference regime sits beside the port's.
REFERENCE_NUMPY = {
    "what": "pollen-robotics/reachy2_symbolic_ik run as is (python 3.10.12, numpy 2.2.6 / OpenBLAS 0.3.29, scipy 1.15.3), one core",
    "measured": "in the build container, NOT on this GPU box (BASELINE.md section 2): Intel Xeon @ 2.10 GHz, 8 cores, 62 GB",
    "harness": "src/benchmark/ik_benchmarks.py:12-33 shape: one fixed pose, tight loop, perf_counter / iterations",
    "config2_solves_per_s_per_core": 1.54e3,        # is_reachable 401 us + get_joints(interval[0]) 248 us
    "config3_solves_per_s_per_core": 0.68e3,        # ControlIK discrete, nb_search_points = 20: 1.48 ms
    "config5_steps_per_s_per_core": 0.9e3,          # continuous on the tests/test_sdk.py:38-63 trajectory: 1.05-1.21 ms per step
    "all_8_cores_independent_processes": {"config2_solves_per_s": 12e3, "config3_solves_per_s": 5.4e3},
    "static": True,
}

# ------------------------------------------------------------------------------------------ synthetic workloads
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


def make_config4_poses(n, seed=20250204, device=0):
    """the config-2 generator, arm id per pose 50/50, l poses = mirror images of r poses
    (pose_l = (x, -y, z; -roll, pitch, -yaw), the G5 rule), so both arms have equal reachability."""
    pos, eul = make_config2_poses(n, seed=seed, device=device)
    arm_id = (np.random.default_rng(seed + 99).uniform(size=n) < 0.5).astype(np.uint8)
    sgn = np.where(arm_id == 1, -1.0, 1.0)
    pos = pos * np.stack([np.ones(n), sgn, np.ones(n)], axis=1)
    eul = eul * np.stack([sgn, np.ones(n), sgn], axis=1)
    return pos, eul, arm_id


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


def make_config5_trajectories(n_traj, n_steps, seed=20250204, device=0):
    """task-space generator shaped like the reference's tests/test_sdk.py:38-63 (x0,y0,z0 =
    0.65,-0.2,0; rpy0 = 0,-pi/2,0; amp 0.35 m / pi/6 rad; freqs 0.6,0.34,0.78,0.18,0.31,0.47; t = k/120 + 11 + phase).
    Returns the goal matrices of every step packed [n_steps, 12, n_traj] on the device."""
    import torch

    dev = torch.device("cuda", device)
    g = torch.Generator(device="cpu").manual_seed(seed)
    phase = (torch.rand(n_traj, generator=g, dtype=torch.float64) * 40.0).to(dev)
    k = torch.arange(n_steps, dtype=torch.float64, device=dev)
    t = (k / 120.0 + 11.0)[:, None] + phase[None, :]
    c0 = [0.65, -0.2, 0.0, 0.0, -np.pi / 2, 0.0]
    amp = [0.35, 0.35, 0.35, np.pi / 6, np.pi / 6, np.pi / 6]
    freq = [0.6, 0.34, 0.78, 0.18, 0.31, 0.47]
    v = [c + a * torch.sin(f * t) for c, a, f in zip(c0, amp, freq)]
    ca, sa, cb, sb, cc, sc = torch.cos(v[3]), torch.sin(v[3]), torch.cos(v[4]), torch.sin(v[4]), torch.cos(v[5]), torch.sin(v[5])
    rows = [cc * cb, cc * sb * sa - sc * ca, cc * sb * ca + sc * sa, sc * cb, sc * sb * sa + cc * ca, sc * sb * ca - cc * sa,
            -sb, cb * sa, cb * ca, v[0], v[1], v[2]]
    return torch.stack(rows, dim=1).contiguous()


# ------------------------------------------------------------------------------------------ CPU baseline leg (+ parity check)
def host_facts():
    """The host the CPU-baseline leg runs on: CPUs this process may use (affinity mask), logical CPUs of the machine, CPU model."""
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.lower().startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    if not model:
        import platform

        model = platform.processor() or platform.machine() or "unknown"
    return {"cores_visible": max(1, visible), "logical_cpus": os.cpu_count() or visible, "cpu_model": model}


def cpu_baseline(config, sample, seconds, gpu=None):
    """Times the CPU checker (oracle/, a C restatement of the reference path = kind "port") on the host cores, on a
    bounded sample of the SAME workload: once as the portable build that travels with the repo (gcc -O2, baseline
    x86-64) and once rebuilt on this host with gcc -O3 -march=native (bit-identical results, FP contraction off in
    both).  Reported baseline, not the target.  `gpu` = the GPU's outputs for the rows of `sample`: since the checker's
    results are at hand they are compared (flags exact, joints <= 1e-6 rad) — with N > 1 the sample rows come from EVERY
    rank's part of the all-gathered arrays."""
    from oracle import oracle as orc

    host = host_facts()
    avail = max(1, min(host["cores_visible"], orc.lib().orc_max_threads()))
    if config in (2, 4):
        pos, eul = np.ascontiguousarray(sample["pos"]), np.ascontiguousarray(sample["eul"])
        m = len(pos)
        arm_id = None if sample.get("arm") is None else np.ascontiguousarray(sample["arm"])
        ar, al = orc.Arm("r_arm", 0.03), orc.Arm("l_arm", 0.03)
        run = lambda nt, L=None: orc.solve_batch(ar, al, pos, eul, arm_id=arm_id, nthreads=nt, L=L)  # noqa: E731
    elif config == 3:
        M = np.ascontiguousarray(sample["M"])
        m = len(M)
        ar, al = orc.Arm("r_arm", -1.01), orc.Arm("l_arm", -1.01)
        run = lambda nt, L=None: orc.control_discrete_batch(ar, al, M, nb_search_points=64, nthreads=nt, L=L)  # noqa: E731
    else:  # config 5: a subsample of the trajectories walked by the checker's state machine (control_ik.py:276-407)
        M = np.ascontiguousarray(sample["M"])                     # [n_steps, n_sub, 4, 4]
        n_steps_s, n_sub = M.shape[:2]
        m = n_steps_s * n_sub                                     # trajectory-steps per pass
        ar = orc.Arm("r_arm", -1.01)
        st0 = np.ascontiguousarray(sample["state0"])              # [n_sub, 11] checker state rows

        def run(nt, L=None):
            return orc.control_continuous_run_batch(ar, st0.copy(), M, first_step_timed_out=True, preferred_theta_self=sample["pref_self"],
                                                    nthreads=nt, L=L)

    def rate(nt, budget, L=None):
        run(nt, L)
        t0 = time.perf_counter()
        passes = 0
        while True:
            run(nt, L)
            passes += 1
            el = time.perf_counter() - t0
            if el >= budget or passes >= 2000:
                return passes * m / el, passes, el

    # the host may expose more logical CPUs than the container's CPU quota: probe a few thread counts briefly,
    # then spend the remaining budget on the best one
    cands = sorted({c for c in (1, 8, 16, 32, 64, 128, avail) if c <= avail})
    probe = {c: rate(c, 0.5)[0] for c in cands}
    best = max(probe, key=probe.get)
    left = max(2.0, seconds - 0.5 * len(cands))
    value, passes, el = rate(best, left / 2)
    base = {
        "value": value, "unit": "steps/s" if config == 5 else "solves/s", "cores": best, "kind": "port", "single_thread": probe[1],
        # the host ("core count and CPU model printed"): `cores` / `threads_used` = the OpenMP threads `value` was
        # measured with (the best of a short probe over thread counts, so two configs of one run may differ in it), `cores_visible` = the
        # CPUs this process may run on (its affinity mask), `cpu_model` from /proc/cpuinfo — the same in every entry of one run
        "threads_used": best, "cores_visible": host["cores_visible"], "logical_cpus": host["logical_cpus"], "cpu_model": host["cpu_model"],
        "threads_probed": {str(c): probe[c] for c in cands},
        "portable_build": {"value": value, "flags": "gcc -O2 -ffp-contract=off (built in the build container, travels with the repo)",
                           "single_thread": probe[1]},
        "sample": f"{m} {'trajectory-steps' if config == 5 else 'poses'} of the workload x {passes} passes ({el:.1f} s) with OpenMP {best} threads "
                  f"(best of {cands}; 1 thread: {probe[1]:.0f} per s)",
    }
    try:
        Ln = orc.native_lib()
        nat1 = rate(1, 0.5, Ln)[0]
        natv, npass, nel = rate(best, left / 2, Ln)
        base["native_build"] = {"value": natv, "flags": "gcc -O3 -march=native -ffp-contract=off (built on this host)",
                                "single_thread": nat1, "passes": npass, "seconds": nel}
        if natv > value:
            base["value"], base["single_thread"] = natv, nat1
            base["sample"] += f"; value = the native build ({natv:.0f} per s; portable build {value:.0f})"
    except Exception as e:  # no gcc on the box: the portable figure stands
        base["native_build"] = {"error": f"{type(e).__name__}: {e}"}
    if gpu is not None:
        ref = run(best)
        np.testing.assert_array_equal(gpu["state"], ref["state"], err_msg="GPU state codes differ from the checker")
        ok = ref["reachable"].astype(bool) if config != 5 else np.ones(ref["reachable"].shape, dtype=bool)  # config 5: every step has joints
        if "reachable" in gpu:
            np.testing.assert_array_equal(gpu["reachable"], ref["reachable"], err_msg="GPU flags differ from the checker")
        err = float(np.max(np.abs(gpu["joints"][ok] - ref["joints"][ok]), initial=0.0))
        assert err < 1e-6, f"GPU joints differ from the checker by {err} rad"
        base["parity_on_sample"] = {"rows": int(m), "reachable_rows": int(ok.sum()), "flags_and_states": "bit-exact",
                                    "max_abs_joint_error_rad": err}
        if config != 5:
            # the workload generators keep poses by the product's own is_reachable kernel (make_config2_poses / make_config3_matrices):
            # the checker must call every kept row of the sample reachable as well, or the workload would be one the GPU chose for itself
            # (config 3: a wrist-reachable pose whose theta search finds nothing reports "limited by shoulder", control_ik.py:452; the
            # codes 1-5 are the ones is_reachable itself gives to a pose it refuses, symbolic_ik.py:159-306)
            kept = int((ref["state"] == 0).sum()) if config in (2, 4) else int(np.isin(ref["state"], (0, 6)).sum())
            assert kept == m, f"the workload filter (HIP is_reachable) kept {m - kept} of {m} sampled rows the checker does not call reachable"
            base["workload_filter"] = {"rows": int(m), "rows_the_checker_calls_reachable": kept,
                                       "what": "generator filter = the HIP is_reachable kernel; asserted here against the CPU checker on the sample"}
    base["reference_numpy"] = REFERENCE_NUMPY
    return base


# ------------------------------------------------------------------------------------------ measurement helpers
def fp64_valu_calibration(hs, n=1 << 20, reps=5):
    """Wave-instructions/s the GPU sustains on pure independent v_fma_f64 (rsik_debug_math op 6: 8 x 2048 per lane),
    measured in the same process right after the timed region: an fp64 reference rate under the power-managed clock."""
    import torch

    a = torch.rand(n, dtype=torch.float64, device=hs.device)
    b = torch.full((n,), 0.999, dtype=torch.float64, device=hs.device)
    hs.debug_math(6, a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        hs.debug_math(6, a, b)
    e1.record()
    torch.cuda.synchronize()
    return (n / 64) * 8 * 2048 * reps / (e0.elapsed_time(e1) * 1e-3)


def clock_ghz(core, real):
    c, r = core.cpu().numpy(), real.cpu().numpy()
    ok = r > 0
    return float(np.median(c[ok] / r[ok]) * 0.1) if ok.any() else None


def sustained_phase(hs, launch_all, load_seconds, monitor_seconds):
    """Replays the step back to back for `load_seconds`, then keeps replaying while a clock monitor (rsik_debug_math
    op 8, one wave per workgroup on a side stream) samples the shader clock for `monitor_seconds`.  Returns the clock the
    chip HOLDS under this kernel's load and the step time measured in that window."""
    import torch

    side = torch.cuda.Stream(device=hs.device)
    t_end = time.perf_counter() + load_seconds
    while time.perf_counter() < t_end:
        for _ in range(128):
            launch_all()
    core, real = hs.clock_monitor(monitor_seconds, 64, side)
    done = torch.cuda.Event()
    done.record(side)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    count = 0
    while not done.query() and count < 4_000_000:
        for _ in range(128):
            launch_all()
        count += 128
    e1.record()
    torch.cuda.synchronize()
    return clock_ghz(core, real), e0.elapsed_time(e1) / max(count, 1)


def first_launches_clock(hs, launch_all, k):
    """Shader clock over the first k back-to-back steps after an idle gap (what a short timed region sees)."""
    import torch

    side = torch.cuda.Stream(device=hs.device)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        launch_all()
    e1.record()
    e1.synchronize()
    est = e0.elapsed_time(e1) * 1e-3
    time.sleep(0.2)  # idle, like the barrier + synchronize in front of a timed region
    core, real = hs.clock_monitor(est * 0.8, 16, side)
    for _ in range(k):
        launch_all()
    torch.cuda.synchronize()
    return clock_ghz(core, real)


def committed_counters(cfg, n, build_id):
    """rocprofv3 PMC results committed under profiles/ (HBM bytes per launch, executed vector instructions per wave).
    Only used when they were collected with THIS build of the library and this workload size."""
    try:
        with open(PROFILE_COUNTERS) as fh:
            doc = json.load(fh)
        t = doc.get(str(cfg))
        if t and t["poses_per_gpu"] == n and doc.get("build_id") == build_id:
            return t
        return {"stale": f"profiles/r06/counters.json was collected with another build or size (library {build_id})"}
    except (OSError, ValueError, KeyError):
        return None


_LIVE_TRAFFIC_FAILED = []  # (a first failure — no rocprofv3, a profiler that hangs — is not repeated for the other configs of the run)


def live_traffic(cfg, n, lib, timeout_s=60.0):
    """HBM traffic of the config's kernel(s) measured NOW, in this run: two child runs of this script under `rocprofv3 --pmc`
    (FETCH_SIZE, then WRITE_SIZE — separate passes, with --kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes; KiB
    units; FETCH_SIZE x 2 on gfx950), a few eager launches each.  Returns {"bytes": per launch (config 5: per pass), ...} or
    {"error": ...}: the committed counters (profiles/) then stand in, labelled."""
    import csv
    import glob
    import shutil
    import statistics
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    if _LIVE_TRAFFIC_FAILED:
        return {"error": "skipped: " + _LIVE_TRAFFIC_FAILED[0]}
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES")) or any(
            k.startswith("ROCPROF") for k in os.environ):
        return {"error": "this process already runs under a profiler: no nested rocprofv3"}
    kernel = {2: "solve_kernel", 3: "control_discrete_kernel", 4: "solve_kernel", 5: "cont_"}[cfg]
    child = [sys.executable, os.path.abspath(__file__), "--config", str(cfg), "--poses", str(n), "--steps", "3", "--warmup", "1", "--launch", "eager",
             "--no-cpu-baseline", "--no-extras", "--no-other-configs", "--no-live-traffic", "--no-steady-state"]
    if cfg == 5:  # (a profiler that serialises dispatches deadlocks on device-word waits: streams tied by events)
        child += ["--phased-variant", "1"]
    if lib:
        child += ["--lib", os.path.abspath(lib)]
    got, t0 = {}, time.perf_counter()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["TMPDIR"] = "/tmp"
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(prefix="rsik_pmc_", dir="/tmp") as d:
            try:
                p = subprocess.run([exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp", env=env,
                                   capture_output=True, text=True, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                _LIVE_TRAFFIC_FAILED.append(f"rocprofv3 --pmc {counter}: no result within {timeout_s:.0f} s")
                return {"error": _LIVE_TRAFFIC_FAILED[0]}
            files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                _LIVE_TRAFFIC_FAILED.append(f"rocprofv3 --pmc {counter} failed (exit {p.returncode}): {(p.stderr or p.stdout)[-300:]}")
                return {"error": _LIVE_TRAFFIC_FAILED[0]}
            per_kernel, starts = {}, 0
            for f in files:
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        name = r["Kernel_Name"]
                        if r["Counter_Name"] != counter or "rsik::" not in name or kernel not in name:
                            continue
                        if cfg != 5 and int(r.get("Grid_Size", r.get("Grid_Size_X", "0"))) != ((n + 255) // 256) * 256:
                            continue  # (the workload generator's filter launches: other grid sizes)
                        short = name.split("rsik::")[1].split("<")[0].split("(")[0]
                        per_kernel.setdefault(short, []).append(float(r["Counter_Value"]))
                        starts += 1 if "cont_init_kernel" in name else 0
            if not per_kernel:
                return {"error": f"rocprofv3 --pmc {counter}: no dispatch of the {kernel} kernel(s) in the counter collection"}
            if cfg == 5:  # the pipeline's kernels of one pass together: launches seen / passes seen (one start-up kernel per pass)
                if not starts:
                    return {"error": "no cont_init_kernel dispatch seen"}
                got[counter] = {k: sum(v) / starts for k, v in per_kernel.items()}
            else:
                got[counter] = {k: statistics.median(v) for k, v in per_kernel.items()}
    by_kernel = {k: 2 * got["FETCH_SIZE"].get(k, 0.0) * 1024 + got["WRITE_SIZE"].get(k, 0.0) * 1024 for k in set(got["FETCH_SIZE"]) | set(got["WRITE_SIZE"])}
    return {"bytes": sum(by_kernel.values()), "by_kernel": by_kernel, "seconds": time.perf_counter() - t0,
            "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes with --kernel-trace, KiB units, FETCH_SIZE x 2 on "
                      "gfx950) around child runs of this script, 4 eager launches each" + (" (config 5: per pass, streams tied by events)" if cfg == 5 else "")}


# ------------------------------------------------------------------------------------------ launcher (N > 1 without torchrun)
def launch_ranks(n_ranks, argv):
    """Starts one child process per GPU and waits for them.  The parent has not imported torch nor touched the GPU (a
    process that has initialised HIP must never be replaced or forked into GPU work on this platform)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile

    procs, logs = [], []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))  # RCCL's own warnings go to the rank's stderr
        log = tempfile.TemporaryFile(mode="w+")  # the rank's stderr: shown (tail) if it fails
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stderr=log))
    rc, alive = 0, set(range(n_ranks))

    def show_tail(r, lines=40):
        logs[r].seek(0)
        tail = logs[r].read().splitlines()[-lines:]
        if tail:
            print(f"---- stderr of rank {r} (last {len(tail)} lines) ----\n" + "\n".join(tail), file=sys.stderr)

    try:
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                    show_tail(r)
                    for q in alive:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:  # exactly the processes started above
            if p.poll() is None:
                p.terminate()
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
        if rc == 0:  # rank 0's warnings (RCCL WARN lines, the libdrm notice) still reach the caller's stderr
            logs[0].seek(0)
            sys.stderr.write(logs[0].read())
        for log in logs:
            log.close()
    return rc


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default 1000; config 5 and N > 1: 20)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5], help="default: 2 at N = 1, 4 at N > 1")
    ap.add_argument("--poses", type=int, default=0, help="poses per GPU (default: the BASELINE size)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--phased-variant", type=int, default=0,
                    help="config 5: RSIK_OPT_CONT_PHASED_VARIANT for the run (1 = streams tied by events also when issued launch by launch: "
                         "what the PMC passes of scripts/profile.sh use — a profiler that serialises dispatches deadlocks on a replayed "
                         "multi-stream graph, and device-word waits are not its business either)")
    ap.add_argument("--no-steady-state", action="store_true",
                    help="skip the second timing (config 5: after 60 more passes, and the other launch form; configs 2-4: after --settle-ms of "
                         "untimed launches) — profiler runs: only the timed form's kernels")
    ap.add_argument("--settle-ms", type=float, default=50.0,
                    help="configs 2-4, N = 1: milliseconds of untimed back-to-back launches ahead of the `steady_state` leg (0 = no such leg)")
    ap.add_argument("--stages", action="store_true",
                    help="add per-stage device figures for configs 2 and 3 (scripts/stage_timers.py, a child process after the timed "
                         "region: launch differences with this library, per-wave stamps with a -DRSIK_TIMELINE_PROBE build)")
    ap.add_argument("--no-valu-calibration", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the cold-HBM, sustained-clock and calibration phases (A/B timing)")
    ap.add_argument("--gather", choices=["step", "final", "none"], default="final",
                    help="N > 1: final (default) = K sharded steps, then ONE all-gather of joints + state inside the timed region — the "
                         "north star's job shape; step = an all-gather inside every timed step; none = never.  The line carries the "
                         "kernel-only, gather-only and both end-to-end figures whichever is timed")
    ap.add_argument("--gather-every-step", action="store_true", help="same as --gather step")
    ap.add_argument("--chunks", type=int, default=4,
                    help="N > 1 with --gather step: stripes per shard; stripe c's all-gather travels while stripe c + 1 is solved")
    ap.add_argument("--launch", choices=["auto", "eager", "graph", "pipelined"], default="auto",
                    help="N = 1: K pre-bound launches from Python (eager) or one replay of a hipGraph holding the K launches; "
                         "auto = graph for K <= 32 and K >= 200, eager between (measured: graphs of 33 ... ~150 nodes replay slower than eager launches).  "
                         "Config 5: auto = pipelined = the K passes issued launch by launch with RSIK_OPT_CONT_GOALS_RESIDENT (the prepare phase of "
                         "pass k + 1 beside the tail of pass k); eager = the same without the overlap; graph = one captured pass replayed K times")
    ap.add_argument("--graph", action="store_true", help="same as --launch graph")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--single-device", action="store_true", help="every rank uses GPU 0 (with --backend gloo: rehearsal on a 1-GPU box)")
    ap.add_argument("--grouped", action="store_true",
                    help="--gpus 1 through the N > 1 code path: a ONE-rank process group (RCCL by default) — group set-up, barriers, the "
                         "all-gather forms, max-over-ranks timing, the multi_gpu block; what a one-GPU box can rehearse of the RCCL path")
    ap.add_argument("--collective-timeout", type=float, default=120.0, help="N > 1: seconds before a stuck collective fails its rank")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="ranks only meet, all-reduce their rank numbers on the CPU and print the line's skeleton (launcher self-test, no GPU)")
    ap.add_argument("--lib", default="", help="alternative build of librsik_hip.so (A/B timing, probe builds)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two short child runs under rocprofv3 --pmc after the timed legs); the "
                         "counters committed under profiles/ are quoted instead when they belong to this build")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default run (N = 1, no --config): do not time configs 3, 4 and 5 after the headline config")
    args = ap.parse_args(argv)
    if args.graph:
        args.launch = "graph"
    if args.gather_every_step:
        args.gather = "step"
    return args


def describe_group(torch, dist, dev, world, rank, args):
    """What the collective library really sees: every rank's device (uuid, PCI bus id, name), host and pid, all-gathered;
    two ranks on one device are refused unless --single-device (a rehearsal) says so.  Returned on every rank."""
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "host": socket.gethostname(), "pid": os.getpid(), "device_index": dev.index, "name": props.name,
          "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": int(getattr(props, "pci_bus_id", -1)),
          "pci_device_id": int(getattr(props, "pci_device_id", -1)), "pci_domain_id": int(getattr(props, "pci_domain_id", -1)),
          "visible": os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES", ""))}
    everyone = [None] * world
    dist.all_gather_object(everyone, me)
    ident = [(e["host"], e["uuid"] or (e["pci_domain_id"], e["pci_bus_id"], e["pci_device_id"], e["device_index"])) for e in everyone]
    distinct = len(set(ident))
    if distinct != world and not args.single_device:
        raise SystemExit(f"bench.py: {world} ranks but only {distinct} distinct devices ({ident}): every rank needs a GPU of its own "
                         "(--single-device rehearses the code path on one)")
    try:
        v = torch.cuda.nccl.version()
        lib_version = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception as e:  # noqa: BLE001
        lib_version = f"unknown ({type(e).__name__})"
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "distinct_devices": distinct,
            "collective_library": ("RCCL " + lib_version) if args.backend == "nccl" else args.backend,
            "ranks": everyone, "collective_timeout_s": args.collective_timeout}


def rendezvous_only(args, world, rank):
    import torch
    import torch.distributed as dist

    dist.init_process_group(backend="gloo")
    t = torch.tensor([float(rank)], dtype=torch.float64)
    dist.all_reduce(t)
    assert float(t[0]) == world * (world - 1) / 2
    if rank == 0:
        print(json.dumps({"metric": "launcher self-test", "n_gpus": world, "rank_sum": float(t[0]), "gpus_flag": args.gpus}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


# ------------------------------------------------------------------------------------------ one rank
def other_config_entry(oc, sub, wall_s):
    """What the default line carries for one of the other BASELINE configs (its own driver-protocol leg: W = 5, K = 20)."""
    r = sub["roofline"]
    e = {
        "workload": sub["config"]["workload"], "metric": sub["metric"], "value": sub["value"], "unit": sub["unit"],
        "steps": sub["steps"], "warmup": sub["warmup"], "ms_per_step": sub["ms_per_step"], "kernel_ms": r["kernel_ms"], "launch": sub["launch"],
        "algorithmic_bytes_per_pose": r["algorithmic_bytes_per_pose"], "achieved_GBs": r["achieved"], "frac": r["frac"],
        "traffic": r.get("traffic"),
        "parity_on_sample": sub.get("cpu_baseline", {}).get("parity_on_sample"),
        "cpu_baseline": {k: sub.get("cpu_baseline", {}).get(k) for k in ("value", "unit", "cores", "threads_used", "cores_visible", "logical_cpus",
                                                                          "cpu_model", "kind", "single_thread", "sample", "workload_filter")},
        "wall_s": wall_s,
    }
    if oc == 5:
        # a bench "step" of config 5 is one PASS (4096 trajectories x 1000 control steps): ms_per_step above is per pass; the
        # kernel time and the traffic are carried in both units, labelled
        e.pop("kernel_ms")
        e.pop("traffic")
        e["ms_per_pass"] = sub["ms_per_step"]
        e["kernel_ms_per_pass"] = r["kernel_ms"] * 1000
        e["kernel_ms_per_control_step_of_4096_trajectories"] = r["kernel_ms"]
        e["traffic_bytes_per_pass"] = (r["traffic"] * 1000) if r.get("traffic") else None
        e["traffic_bytes_per_control_step_of_4096_trajectories"] = r.get("traffic")
        e["frac_at_286_bytes_state_round_trip_per_step"] = r.get("frac_at_286_bytes_state_round_trip_per_step")
        e["value_is"] = sub.get("value_is")
        e["run_forms_seen"] = sub.get("run_forms_seen")
    e["steady_state"] = sub.get("steady_state")
    return e


def run_default_protocol(make_leg, headline_cfg, others=(3, 4, 5), log=None):
    """The default run (N = 1, no --config): the headline config and the other BASELINE configs in ONE protocol —
        1. every config's synthetic workload is generated and made resident (no timing yet),
        2. the GPU legs back to back: the headline's warm-up + K timed steps (+ its extras), then each other config's W = 5 / K = 20,
        3. only then the CPU-baseline legs (seconds of host work during which the GPU idles and clocks down),
    so that no config is timed on a chip that sat idle behind another one's CPU leg.  `make_leg(cfg)` returns a generator that
    yields "ready" after its set-up and "timed" after its GPU legs and returns its line (bench._run); a config other than the
    headline that fails is reported in its entry and dropped.  Returns (headline line, {cfg: line | {"error": ...}}, order)."""
    order, t_start = [], time.perf_counter()

    def note(what, cfg):
        order.append([what, cfg, round(time.perf_counter() - t_start, 3)])
        if log:
            log(what, cfg)

    legs, failed, wall = {}, {}, {}

    def advance(cfg, want):
        t0 = time.perf_counter()
        try:
            got = next(legs[cfg])
            assert got == want, (got, want)
        except Exception as e:  # (StopIteration included: a leg that ends early is an error too)
            if cfg == headline_cfg:
                raise
            failed[cfg] = {"error": f"{type(e).__name__}: {e}"}
            legs.pop(cfg, None)
        wall[cfg] = wall.get(cfg, 0.0) + time.perf_counter() - t0

    for cfg in (headline_cfg,) + tuple(c for c in others if c != headline_cfg):
        legs[cfg] = make_leg(cfg)
        advance(cfg, "ready")
        note("setup", cfg)
    for cfg in list(legs):
        advance(cfg, "timed")
        note("gpu", cfg)
    lines = {}
    for cfg in list(legs):
        t0 = time.perf_counter()
        try:
            while True:
                next(legs[cfg])
        except StopIteration as stop:
            lines[cfg] = stop.value
        except Exception as e:
            if cfg == headline_cfg:
                raise
            failed[cfg] = {"error": f"{type(e).__name__}: {e}"}
        wall[cfg] = wall.get(cfg, 0.0) + time.perf_counter() - t0
        note("cpu", cfg)
    out = {}
    for cfg in others:
        if cfg == headline_cfg:
            continue
        out[cfg] = failed[cfg] if cfg in failed else other_config_entry(cfg, lines[cfg], wall.get(cfg, 0.0))
    return lines[headline_cfg], out, order


def _drain(gen):
    try:
        while True:
            next(gen)
    except StopIteration as stop:
        return stop.value


def main(argv=None, return_line=False):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    in_group = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if not in_group and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, argv))
    if args.grouped and not in_group:  # a one-rank group in this process
        import socket

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.config == 0 and not args.no_other_configs and not args.rendezvous_only and not args.grouped:
        others_argv = ["--steps", "20", "--warmup", "5", "--no-extras", "--no-valu-calibration", "--no-other-configs", "--cpu-seconds", "2"]
        if args.lib:
            others_argv += ["--lib", args.lib]
        line, others, order = run_default_protocol(
            lambda cfg: _run(argv + ["--config", "2"]) if cfg == 2 else _run(["--config", str(cfg)] + others_argv), 2)
        line["other_configs"] = {str(k): v for k, v in others.items()}
        line["protocol"] = {"order": order,
                            "what": "[phase, config, seconds since start]: every workload resident first, then the GPU legs back to back "
                                    "(headline first), then the CPU-baseline legs"}
    else:
        line = _drain(_run(argv))
    if return_line:
        return line
    if line is not None:
        if args.stages and world == 1:
            import subprocess

            sp = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "stage_timers.py")],
                                capture_output=True, text=True, timeout=900)
            got = [ln for ln in sp.stdout.splitlines() if ln.startswith("{")]
            line["stages"] = json.loads(got[-1]) if sp.returncode == 0 and got else {"error": (sp.stderr or sp.stdout)[-600:]}
        print(json.dumps(line), flush=True)


def _run(argv):
    """One config on one rank, as a generator: yields "ready" once the workload is resident and the launches are bound, "timed" once
    every GPU leg is done, and returns the line (rank 0; None on the other ranks) after the CPU-baseline leg."""
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"(or without torchrun: bench.py starts the ranks itself)")
    if args.rendezvous_only:
        rendezvous_only(args, world, rank)
        return None
    grouped = world > 1 or args.grouped  # (the N > 1 code path; --grouped: with one rank)

    cfg = args.config or (4 if grouped else 2)
    if not args.steps:
        args.steps = 20 if (cfg == 5 or grouped) else 1000
    if args.lib:
        from reachy2_symbolic_ik_amd import _abi

        if _abi.LIB_PATH != os.path.abspath(args.lib):  # (the legs of a default run share the process and the library)
            _abi.use_library(args.lib)

    import torch

    from reachy2_symbolic_ik_amd import ControlIK, SymbolicIK
    from reachy2_symbolic_ik_amd.distributed import ShardedBuffers, ShardPlan, gather_stripe

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if grouped:
        import torch.distributed as dist

        import datetime

        # a collective that does not complete fails the rank (and with it the launcher) instead of hanging the box
        timeout = datetime.timedelta(seconds=args.collective_timeout)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev, timeout=timeout)
        else:
            dist.init_process_group(backend=args.backend, timeout=timeout)
        group_info = describe_group(torch, dist, dev, world, rank, args)
    n = args.poses or {2: 1 << 20, 3: 1 << 18, 4: 1 << 20, 5: 4096}[cfg]
    gather_mode = args.gather if (grouped and cfg != 5) else "none"
    chunks = max(1, args.chunks) if gather_mode == "step" else 1
    if n % chunks:
        raise SystemExit(f"--poses {n} must be a multiple of --chunks {chunks}")
    rpp = n // chunks  # rows per piece
    plan = ShardPlan(world * n, world, chunks)
    assert plan.rows_per_piece == rpp
    form0 = (plan, chunks, rpp)
    f64, u8 = torch.float64, torch.uint8

    def local_to_global(i):  # row i of this rank's shard -> row of the all-gathered array (block-cyclic over stripes)
        return (i // rpp) * plan.stripe_rows + rank * rpp + (i % rpp)

    # ---- synthetic inputs, resident in HBM before the timed region; outputs; one pre-bound launch per stripe
    gathered_names = ("joints", "state")
    bufs = None
    sample_local = {}
    hs = None
    if cfg in (2, 4):
        if cfg == 2:
            base_n = min(n, 1 << 20)
            pos, eul = make_config2_poses(base_n, seed=20250204 + rank, device=local_rank)
            if n > base_n:  # size sweeps beyond the BASELINE size reuse the same reachable poses (timing only)
                reps = (n + base_n - 1) // base_n
                pos, eul = np.tile(pos, (reps, 1))[:n], np.tile(eul, (reps, 1))[:n]
            arm_id = None
            solver_obj = _quiet(SymbolicIK, "r_arm", device=local_rank)
            hs = solver_obj.solver
            workload = f"config2: r_arm is_reachable + theta_to_joints_func(interval[0]), {n} random reachable poses per GPU"
            kernel_name = "solve_kernel"
        else:
            from reachy2_symbolic_ik_amd import DualArmIK

            pos, eul, arm_id = make_config4_poses(n, seed=20250204 + rank, device=local_rank)
            solver_obj = _quiet(DualArmIK, device=local_rank)
            hs = solver_obj.solver
            workload = f"config4: r_arm + l_arm mixed (per-pose arm byte), {n} reachable poses per GPU, theta = interval[0]"
            kernel_name = "solve_kernel<mixed>"
        soa = torch.as_tensor(np.ascontiguousarray(np.concatenate([pos.T, eul.T], axis=0))).to(dev)
        arm_t = None if arm_id is None else torch.as_tensor(arm_id).to(dev)
        sample_local = {"pos": pos, "eul": eul, "arm": arm_id}

        def make_set(soa_t, arm_tt, shared=None, form=None):
            """Output buffers + one pre-bound rsik_solve launch per stripe for one resident copy of the inputs (`form`: another
            partition of the shard into stripes than the run's own — the other gather form's timed leg)."""
            plan, chunks, rpp = form or form0
            if grouped:
                sb = shared or ShardedBuffers(plan, rank, {"joints": ((7,), f64), "state": ((), u8)}, dev)
                o = {"interval": torch.empty((n, 2), dtype=f64, device=dev), "reachable": torch.empty((n,), dtype=u8, device=dev)}
            else:
                sb = None
                o = {"joints": torch.empty((n, 7), dtype=f64, device=dev), "interval": torch.empty((n, 2), dtype=f64, device=dev),
                     "reachable": torch.empty((n,), dtype=u8, device=dev), "state": torch.empty((n,), dtype=u8, device=dev)}
            launches, keep = [], []
            for c in range(chunks):
                a, b = c * rpp, (c + 1) * rpp
                if sb is not None:
                    pv = sb.piece_views(c)
                    po = {"joints": pv["joints"], "state": pv["state"], "interval": o["interval"][a:b], "reachable": o["reachable"][a:b]}
                else:
                    po = {k: v[a:b] for k, v in o.items()}
                if arm_tt is None:
                    p = solver_obj.solve_batch(soa_t[:, a:b], want_elbow=False, out=po, plan_only=True)
                else:
                    p = solver_obj.solve_batch(arm_tt[a:b], soa_t[:, a:b], want_elbow=False, out=po, plan_only=True)
                launches.append(p["launch"])
                keep.append(p)
            return {"out": o, "bufs": sb, "launches": launches, "keep": keep, "inputs": (soa_t, arm_tt)}

        main_set = make_set(soa, arm_t)
        out, bufs, launches = main_set["out"], main_set["bufs"], main_set["launches"]
        units = n
    elif cfg == 3:
        from reachy2_symbolic_ik_amd.control_ik import matrices_to_m12_soa

        M = make_config3_matrices(n, seed=20250204 + rank, device=local_rank)
        sample_local = {"M": M}
        ctrl = _quiet(ControlIK, urdf_path=URDF, device=local_rank)
        ctrl.nb_search_points = 64
        hs = ctrl._solver
        m12 = matrices_to_m12_soa(M, dev)

        def make_set(m12_t, _unused=None, shared=None, form=None):
            plan, chunks, rpp = form or form0
            if grouped:
                sb = shared or ShardedBuffers(plan, rank, {"joints": ((7,), f64), "state": ((), u8)}, dev)
                o = {"reachable": torch.empty((n,), dtype=u8, device=dev), "emergency": torch.empty((n,), dtype=u8, device=dev)}
            else:
                sb = None
                o = {"joints": torch.empty((n, 7), dtype=f64, device=dev), "reachable": torch.empty((n,), dtype=u8, device=dev),
                     "state": torch.empty((n,), dtype=u8, device=dev), "emergency": torch.empty((n,), dtype=u8, device=dev)}
            launches, keep = [], []
            for c in range(chunks):
                a, b = c * rpp, (c + 1) * rpp
                if sb is not None:
                    pv = sb.piece_views(c)
                    po = {"joints": pv["joints"], "state": pv["state"], "reachable": o["reachable"][a:b], "emergency": o["emergency"][a:b]}
                else:
                    po = {k: v[a:b] for k, v in o.items()}
                p = ctrl.symbolic_inverse_kinematics_batch("r_arm", m12_t[:, a:b], out=po, plan_only=True)
                launches.append(p["launch"])
                keep.append(p)
            return {"out": o, "bufs": sb, "launches": launches, "keep": keep, "inputs": (m12_t, None)}

        main_set = make_set(m12)
        out, bufs, launches = main_set["out"], main_set["bufs"], main_set["launches"]
        workload = f"config3: r_arm ControlIK discrete, 64-point theta sweep, {n} wrist-reachable goal matrices per GPU"
        kernel_name = "control_discrete_kernel"
        units = n
    else:
        n_steps, n_traj = 1000, n
        traj = make_config5_trajectories(n_traj, n_steps, seed=20250204 + rank, device=local_rank)
        ctrl = _quiet(ControlIK, urdf_path=URDF, device=local_rank)
        hs = ctrl._solver
        if args.phased_variant:
            from reachy2_symbolic_ik_amd import _abi as _A

            hs.set_option(_A.OPT_CONT_PHASED_VARIANT, args.phased_variant)
        cont = ctrl.new_continuous_state("r_arm", n_traj)
        cont0 = cont.clone()
        out = {"joints": torch.empty((n_steps, n_traj, 7), dtype=f64, device=dev),
               "reachable": torch.empty((n_steps, n_traj), dtype=u8, device=dev),
               "state": torch.empty((n_steps, n_traj), dtype=u8, device=dev)}

        # (round 6) consecutive passes may overlap: the goals are resident for the whole run and nothing but the passes themselves
        # touches `out` between them — the promise RSIK_OPT_CONT_GOALS_RESIDENT asks for (include/rsik.h).  The passes stay what they
        # were: each resets the trajectory state, re-initialises every trajectory and writes every output row.
        pipelined = [args.launch in ("auto", "pipelined") and not grouped and not args.phased_variant]
        forms_seen = {}
        # overlapping passes write two sets of output buffers in turn, as a caller that consumes pass k while pass k + 1 runs has to
        # (the promise covers a run's output rows too); every other form writes `out`.  `last_out`: what the last pass issued wrote.
        out_sets = [out, {k: torch.empty_like(v) for k, v in out.items()}] if pipelined[0] else [out]
        last_out, passes_issued = [out], [0]

        def one_pass(stream=None):  # one bench "step" = one 1000-step pass over all trajectories
            cont.copy_(cont0)
            o = out_sets[passes_issued[0] % len(out_sets)] if pipelined[0] else out
            passes_issued[0] += 1
            last_out[0] = o
            r = ctrl.run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=o,
                                                 goals_resident=pipelined[0])
            forms_seen[r.run_form_name] = forms_seen.get(r.run_form_name, 0) + 1

        # the captured form lives on a context of its own: a context a hipGraph points into does not overlap its launch-by-launch runs
        # (the library cannot see replays)
        graph_ctx = {}

        def one_pass_for_capture(stream=None):
            if "ctrl" not in graph_ctx:
                graph_ctx["ctrl"] = _quiet(ControlIK, urdf_path=URDF, device=local_rank)
                graph_ctx["ctrl"]._upload_arms()
                graph_ctx["ctrl"]._solver.control_continuous_reserve(n_traj, n_steps)
            cont.copy_(cont0)
            last_out[0] = out
            graph_ctx["ctrl"].run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=out)

        launches = [one_pass]
        workload = (f"config5: ControlIK continuous, {n_traj} trajectories x {n_steps} steps per pass, trajectory state carried "
                    "on chip across the steps of a pass and in HBM across passes / launches")
        kernel_name = "cont_{init,prepare,theta,joints,chain}_kernel (the trajectory pipeline; kernel_ms = one pass / 1000 steps)"
        units = n * n_steps  # trajectory-steps per bench step

    def launch_all(stream=None):
        for f in launches:
            f(stream)

    def gather_all(async_op=True):
        works = []
        for c in range(chunks):
            works += gather_stripe(bufs, c, gathered_names, async_op=async_op)
        return works

    def step():
        if gather_mode == "step":
            works = []
            for c, f in enumerate(launches):
                f()
                works += gather_stripe(bufs, c, gathered_names, async_op=True)
            for w in works:  # the compute stream waits for the collectives (the next step rewrites their buffers)
                w.wait()
        else:
            launch_all()

    def fence():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    def progress(what):  # N > 1: a line per leg on rank 0's stderr (a rehearsal over gloo at full size is minutes of silence otherwise)
        if grouped and rank == 0:
            print(f"[bench] {time.strftime('%H:%M:%S')} {what}", file=sys.stderr, flush=True)

    progress(f"config {cfg}: {world} ranks, {n} rows per rank resident, launches bound; warm-up")
    yield "ready"
    for _ in range(args.warmup):
        step()
    fence()
    progress(f"warm-up done; timing {args.steps} steps (--gather {gather_mode})")

    # ---- timed region: exactly K steps, bracketed by barrier + synchronize on both sides
    # auto: replay the K launches from a hipGraph where that is the faster way to issue them.  Measured per step on 1 M-pose
    # batches (eager / graph, us): K = 10: 35.6 / 34.5, 20: 33.7 / 33.1, 32: 33.2 / 33.0, 33: 33.2 / 34.0, 50: 34.1 / 38.6,
    # 64: 33.8 / 39.1, 100: 36.9 / 38.7, 200: 37.3 / 34.1, 1000: - / 31.1 — a graph of 33 ... ~150 kernel nodes replays
    # slower than eager launches, shorter and longer ones faster.
    # Config 5 captured: one pass is 9 kernels on four streams tied by events; the graph holds ONE pass (the side streams join the
    # capture through the events the run records) and is replayed K times, same bits (scripts/c5_graph.py).
    # Round 4: issued eagerly the pipeline ties its launches with stream value waits (hipStreamWriteValue32 / WaitValue32, ~4 us an edge
    # against ~11 for an event) and starts each block's theta walk before the previous block's joints fill the chip, which a capture
    # cannot hold — 0.41 -> 0.37-0.39 ms eager; the replayed graph is still 1-3 % ahead and steadier (same box, alternating processes,
    # W = 5 / K = 20: 0.380-0.386 against 0.387-0.391; after 60 more passes 0.363-0.373 against 0.369-0.390), so `auto` replays;
    # both forms are timed below (`steady_state.launch_forms_ms`).
    use_graph = not grouped and (args.launch == "graph" or (args.launch == "auto" and cfg != 5 and (args.steps <= 32 or args.steps >= 200)))
    graph = None
    graph_replays = 1
    if use_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            if cfg == 5:
                one_pass_for_capture()  # (creates the capture's own context, its workspace, streams and events: nothing of that inside a capture)
                fence()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                cs = torch.cuda.current_stream(dev).cuda_stream  # the pre-bound launches go to the capture stream
                if cfg == 5:
                    one_pass_for_capture(cs)
                else:
                    for _ in range(args.steps):
                        launch_all(cs)
            graph_replays = args.steps if cfg == 5 else 1
            # untimed: the captured graph replayed once (configs 2-4: that is K launches); config 5's graph holds ONE pass, so
            # the W warm-up steps are repeated as replays — the thing that is timed (the eager warm-up above ran another form)
            for _ in range(max(1, args.warmup) if cfg == 5 else 1):
                graph.replay()
            fence()
        except Exception as e:  # capture refused: time K eager launches instead (reported in "launch")
            print(f"bench.py: hipGraph capture failed ({type(e).__name__}: {e}); falling back to eager launches", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record()  # events are created lazily at their first record: not inside the timed region
    fence()
    t0 = time.perf_counter()
    e0.record()
    if graph is not None:
        for _ in range(graph_replays):
            graph.replay()  # exactly K launches of the hot path
    else:
        for _ in range(args.steps):
            step()
        if gather_mode == "final":  # the job's one exchange: the last result travels to every rank, inside the timed region
            for w in gather_all():
                w.wait()
    e1.record()
    fence()
    elapsed = time.perf_counter() - t0
    step_ms_events = e0.elapsed_time(e1) / args.steps
    progress(f"timed region done: {elapsed * 1e3:.1f} ms")

    # ---- config 5: the same K passes again after 60 more untimed ones (the clock has settled: the driver's W = 5 protocol above is the
    # headline, this is the steady state), in the form timed above and in the other one (eager <-> one captured pass replayed)
    steady = None
    if cfg == 5 and not grouped and not args.no_steady_state:
        # (three windows of K passes, the median reported: issued launch by launch a pass is ~18 launches and ~30 stream operations, and
        # the HIP runtime's host side hiccups now and then — 1-2 ms around the 1100th launch of a process, 38-49 ms every ~3800 launches,
        # scripts/probes/c5_pass_sequence.py — which one window of 20 passes either contains or not)
        windows_ms = {}

        def timed_passes(fn, w, k, name=None):
            for _ in range(w):
                fn()
            got = []
            for _ in range(3):
                fence()
                e0.record()
                for _ in range(k):
                    fn()
                e1.record()
                fence()
                got.append(e0.elapsed_time(e1) / k)
            if name:
                windows_ms[name] = got
            return float(np.median(got))

        mine = "graph" if graph is not None else ("pipelined" if pipelined[0] else "eager")
        was_pipelined = pipelined[0]

        def form_fn(name):
            if name == "graph":
                if graph is not None:
                    return graph.replay
                one_pass_for_capture()
                fence()
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, capture_error_mode="thread_local"):
                    one_pass_for_capture(torch.cuda.current_stream(dev).cuda_stream)
                graph_ctx["g2"] = g2
                return g2.replay

            def launch_by_launch():
                pipelined[0] = name == "pipelined"
                step()

            return launch_by_launch

        forms = {}
        for name in [mine] + [f for f in ("pipelined", "eager", "graph") if f != mine and not (f == "pipelined" and args.phased_variant)]:
            try:
                forms[name] = timed_passes(form_fn(name), 60, args.steps, name)
            except Exception as e:  # the other forms are information only
                if name == mine:
                    raise
                forms[name] = None
                forms[name + "_error"] = f"{type(e).__name__}: {e}"
        # a pass on its own: the device idle before it, HIP events around it (launch by launch, no overlap)
        pipelined[0] = False
        iso = []
        for _ in range(max(3, min(args.steps, 10))):
            fence()
            e0.record()
            step()
            e1.record()
            fence()
            iso.append(e0.elapsed_time(e1))
        pipelined[0] = was_pipelined
        steady = {"after_untimed_passes": args.warmup + args.steps + 60, "steps": args.steps, "ms_per_step": forms[mine],
                  "value": units / (forms[mine] * 1e-3), "unit": "steps/s", "launch": mine,
                  "frac": BYTES_PER_POSE[5] * units / (forms[mine] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  "launch_forms_ms": forms,
                  "launch_forms_windows_ms": windows_ms,
                  "windows": "every form: 60 untimed passes, then three windows of K timed passes (HIP events, a synchronisation between windows); the median is quoted",
                  "launch_forms": {"pipelined": "K passes launch by launch with RSIK_OPT_CONT_GOALS_RESIDENT: the prepare phase of pass k + 1 runs beside "
                                                "the last chain kernel of pass k and its own start-up search (everything else of pass k + 1 waits for pass k's "
                                                "end: the trajectory state); two sets of output buffers in turn",
                                   "eager": "K passes launch by launch, every pass's four streams meet at its start and at its end (no overlap between passes)",
                                   "graph": "one captured pass (two blocks, events) replayed K times: replays of one graph do not overlap"},
                  "isolated_pass_ms": float(np.median(iso)),
                  "isolated_pass_what": f"median of {len(iso)} passes issued launch by launch, each with the device idle before it (a synchronisation "
                                        "between passes: the host's issue latency is inside the figure)"}
        # the results checked below are the timed form's: one more pass in that form
        (graph.replay if graph is not None else step)()
        fence()

    # ---- configs 2-4 on one GPU: the same K steps again behind >= `--settle-ms` of untimed launches of the same kind (round 6: the
    # first ~25 launches of a process run below the sustained clock — docs/experiments.md A.5 — so the driver's W = 5 / K = 20 figure of a
    # 14 us kernel reads 10 % under what profiles/ and a longer run show; this leg lets the driver's own run witness the settled figure,
    # next to the headline it does not replace)
    if cfg != 5 and not grouped and not args.no_steady_state and args.settle_ms > 0:
        per_round_ms = max(step_ms_events * args.steps, 1e-3)
        rounds = max(1, int(np.ceil(args.settle_ms / per_round_ms)))

        def k_steps():
            if graph is not None:
                graph.replay()
            else:
                for _ in range(args.steps):
                    step()

        fence()
        for _ in range(rounds):
            k_steps()
        e0.record()
        k_steps()
        e1.record()
        fence()
        s_ms = e0.elapsed_time(e1) / args.steps
        steady = {"after_untimed_launches": args.warmup + args.steps * (1 + rounds) + (args.steps if graph is not None else 0),
                  "untimed_ms_target": args.settle_ms, "untimed_rounds_of_K": rounds, "steps": args.steps, "ms_per_step": s_ms, "kernel_ms": s_ms,
                  "value": units / (s_ms * 1e-3), "unit": "solves/s", "launch": "graph" if graph is not None else "eager",
                  "achieved_GBs": BYTES_PER_POSE[cfg] * units / (s_ms * 1e-3) / 1e9,
                  "frac": BYTES_PER_POSE[cfg] * units / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  "what": f"the same K = {args.steps} launches timed again (HIP events, one pair) directly behind {rounds} untimed rounds of K "
                          f"(>= {args.settle_ms:g} ms of back-to-back launches): the kernel at the clock the chip settles to"}

    # ---- N > 1: the OTHER gather form, timed the same way (round 6: `value` is whichever form --gather names — since round 5 the
    # north star's job shape, K sharded steps + ONE all-gather, by default; rounds 1-4 timed an all-gather inside every step — and both
    # are measured in every run, each in its own overlapped form, so that lines of different rounds can be compared like for like)
    other_form = None
    if grouped and gather_mode != "none" and cfg != 5:
        o_mode = "step" if gather_mode == "final" else "final"
        o_chunks = max(1, args.chunks) if o_mode == "step" else 1
        if n % o_chunks == 0:
            o_form = (ShardPlan(world * n, world, o_chunks), o_chunks, n // o_chunks)
            o_set = make_set(main_set["inputs"][0], main_set["inputs"][1], form=o_form)

            def o_step():
                if o_mode == "step":
                    works = []
                    for c, f in enumerate(o_set["launches"]):
                        f()
                        works += gather_stripe(o_set["bufs"], c, gathered_names, async_op=True)
                    for w in works:
                        w.wait()
                else:
                    for f in o_set["launches"]:
                        f()

            for _ in range(max(1, min(args.warmup, 3))):
                o_step()
            fence()
            t_o = time.perf_counter()
            for _ in range(args.steps):
                o_step()
            if o_mode == "final":
                for c in range(o_chunks):
                    for w in gather_stripe(o_set["bufs"], c, gathered_names, async_op=True):
                        w.wait()
            fence()
            o_elapsed = time.perf_counter() - t_o
            t = torch.tensor([o_elapsed], dtype=f64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            other_form = {"mode": o_mode, "chunks": o_chunks, "elapsed_s": float(t[0])}
            del o_set
            progress(f"the other gather form (--gather {o_mode}) timed: {other_form['elapsed_s'] * 1e3:.1f} ms")

    # ---- kernel-only and gather-only legs (N > 1), each on its own: same buffers, same launches
    kernel_ms, gather_ms = step_ms_events, 0.0
    if grouped:
        k2 = max(5, min(args.steps, 50))
        fence()
        e0.record()
        for _ in range(k2):
            launch_all()
        e1.record()
        fence()
        kernel_ms = e0.elapsed_time(e1) / k2
        progress(f"kernel-only leg done: {kernel_ms * 1e3:.1f} us per step")
        if gather_mode != "none":
            for w in gather_all():
                w.wait()
            fence()
            e0.record()
            for _ in range(k2):
                for w in gather_all():
                    w.wait()
            e1.record()
            fence()
            gather_ms = e0.elapsed_time(e1) / k2
            progress(f"gather-only leg done: {gather_ms:.1f} ms per all-gather")
            # every rank's rows must have arrived where the partition says: a checksum of checksums over the ranks
            launch_all()
            for w in gather_all():
                w.wait()
            torch.cuda.synchronize()
            mine = torch.stack([bufs.piece_views(c)["joints"].view(torch.int64).sum() for c in range(chunks)]).sum().reshape(1)
            sums = torch.empty((world,), dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(sums, mine)
            for r in range(world):
                got = sum(int(bufs.full["joints"][plan.piece(r, c)[0]: plan.piece(r, c)[1]].view(torch.int64).sum()) for c in range(chunks))
                got = (got + (1 << 63)) % (1 << 64) - (1 << 63)
                assert got == int(sums[r]), f"rank {rank}: rows gathered from rank {r} do not match what it solved"
        t = torch.tensor([elapsed, kernel_ms, gather_ms, step_ms_events], dtype=f64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms, gather_ms, step_ms_events = (float(v) for v in t)

    # sanity: the timed outputs are real results
    if cfg in (2, 4):
        n_ok = int(out["reachable"].sum().item())
        assert n_ok == n, f"{n - n_ok} poses of the reachable workload came back unreachable"
        jj = bufs.full["joints"] if bufs is not None else out["joints"]
        assert bool(torch.isfinite(jj).all()), "non-finite joints"
    if cfg == 5:  # every step of every trajectory produced joints; the reachable / fallback mix is the generator's
        out = last_out[0]  # (the set the last pass wrote: overlapping passes take two in turn)
        assert bool(torch.isfinite(out["joints"]).all()), "non-finite joints in the trajectory batch"
        frac_ok = float(out["reachable"].to(torch.float32).mean().item())
        assert 0.05 < frac_ok < 0.95, f"unexpected reachable fraction {frac_ok}"

    # ---- the rows the CPU-baseline leg re-solves, and the GPU's results for them
    sample, gpu_rows = None, None
    if not args.no_cpu_baseline and cfg == 5 and not grouped:
        # 64 of the trajectories (evenly spread), all their steps: what the CPU leg re-walks
        sub = torch.arange(0, n, max(1, n // 64), device=dev)[:64]
        m12 = traj[:, :, sub].cpu().numpy()                         # [n_steps, 12, n_sub]
        Ms = np.tile(np.eye(4), (m12.shape[0], m12.shape[2], 1, 1))
        Ms[:, :, :3, :3] = np.moveaxis(m12[:, :9, :], 1, 2).reshape(m12.shape[0], m12.shape[2], 3, 3)
        Ms[:, :, :3, 3] = np.moveaxis(m12[:, 9:, :], 1, 2)
        st0 = np.zeros((len(sub), 11))
        c0 = cont0[:, sub].cpu().numpy()
        st0[:, 0], st0[:, 1:8], st0[:, 8], st0[:, 9], st0[:, 10] = c0[0], c0[1:8].T, c0[8], c0[9], c0[10]
        sample = {"M": Ms, "state0": st0, "pref_self": float(ctrl.preferred_theta["r_arm"])}
        gpu_rows = {k: out[k][:, sub].cpu().numpy() for k in ("joints", "state", "reachable")}
    elif not args.no_cpu_baseline and cfg != 5:
        m = min(n, 1 << 18 if cfg in (2, 4) else 1 << 17)
        if not grouped:
            sample = {k: (None if v is None else v[:m]) for k, v in sample_local.items()}
            gpu_rows = {k: out[k][:m].cpu().numpy() for k in ("joints", "state", "reachable")}
        elif gather_mode != "none":
            # the inputs of every rank travel to rank 0 (untimed) so that the checked rows come from all parts of the
            # gathered arrays: m / world consecutive rows of every rank's first stripe
            per = max(1, min(m // world, rpp))
            keys = [k for k, v in sample_local.items() if v is not None]
            sample = {}
            for k in keys:
                loc = torch.as_tensor(np.ascontiguousarray(sample_local[k][:per])).to(dev)
                full = torch.empty((world * per,) + tuple(loc.shape[1:]), dtype=loc.dtype, device=dev)
                dist.all_gather_into_tensor(full, loc)
                sample[k] = full.cpu().numpy()
            if "arm" in sample_local and sample_local["arm"] is None:
                sample["arm"] = None
            rows = torch.as_tensor(np.concatenate([plan.piece(r, 0)[0] + np.arange(per) for r in range(world)])).to(dev)
            gpu_rows = {k: bufs.full[k][rows].cpu().numpy() for k in gathered_names}

    line = None
    if rank == 0:
        total = units * world * args.steps
        value = total / elapsed
        bpp = BYTES_PER_POSE[cfg]
        achieved = bpp * units / (kernel_ms * 1e-3) / 1e9
        collective = "none"
        if grouped:
            collective = {"step": f"RCCL all-gather of joints [n,7] f64 + state u8 inside every step, {chunks} stripes per shard, "
                                  "stripe c in flight while stripe c + 1 is solved",
                          "final": "none in the K sharded steps; ONE RCCL all-gather of joints [n,7] f64 + state u8 after the last of them, "
                                   "inside the timed region",
                          "none": "none"}[gather_mode]
            if args.backend != "nccl" and gather_mode != "none":
                collective += f" [backend {args.backend}: rehearsal, not RCCL]"
        line = {
            "metric": "IK solves/sec (7-DoF r_arm, batched poses)",
            "value": value,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "launch": (((("pipelined: " if pipelined[0] else "eager: ") + "every pass issues the pipeline's kernels on four streams, tied by stream value "
                         "waits (hipStreamWriteValue32 / hipStreamWaitValue32)" + ("; consecutive passes overlap (RSIK_OPT_CONT_GOALS_RESIDENT): the prepare phase of "
                         "pass k + 1 beside the tail of pass k" if pipelined[0] else "")) if graph is None else
                        "K replays of a hipGraph holding one pass (two blocks under capture: 9 kernels on four streams, tied by events)") if cfg == 5
                       else ("eager" if graph is None else "hipGraph replay of K captured launches")),
            "config": {"workload": workload, "poses_per_gpu": n,
                       "theta_policy": {2: "interval[0]", 3: "discrete sweep nb=64", 4: "interval[0]", 5: "continuous, d_theta_max=0.01"}[cfg],
                       "collective": collective},
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
                "kernel_timing": ("HIP events on the launch stream around the K timed steps (one pair)" if not grouped else
                                  "HIP events around a kernel-only replay of the same launches, no collective in flight"),
                "kernel_only_solves_per_s_per_gpu": units / (kernel_ms * 1e-3),
                "note": "fp64 VALU-bound path (DESIGN.md section 4): the HBM fraction is reported as the contract asks; `compute` holds the binding limit",
            },
        }
        if grouped:
            recv = GATHER_BYTES_PER_POSE * n * (world - 1)
            links = min(world - 1, XGMI_LINKS)
            line["multi_gpu"] = {
                "kernel_only_solves_per_s": units * world / (kernel_ms * 1e-3),
                "gather_only_ms": gather_ms if gather_mode != "none" else None,
                "gather_only_solves_per_s": (units * world / (gather_ms * 1e-3)) if gather_ms > 0 else None,
                "end_to_end_ms_per_step": elapsed / args.steps * 1e3,
                "end_to_end_solves_per_s": value,
                "value_is": {"step": "end to end, an all-gather inside every step: n x world x K / (K x (solve + all-gather, overlapped stripe by stripe))",
                             "final": "end to end, the north star's job shape: n x world x K / (K sharded solve steps + ONE all-gather of the final joints + state)",
                             "none": "kernel-only steps (no collective anywhere)"}[gather_mode],
                "gather_bytes_received_per_gpu": recv,
                "xgmi": {"achieved": (recv / (gather_ms * 1e-3) / 1e9) if gather_ms > 0 else None, "unit": "GB/s received per GPU",
                         "peak": links * XGMI_LINK_GBS, "links_usable": links,
                         "frac": (recv / (gather_ms * 1e-3) / 1e9 / (links * XGMI_LINK_GBS)) if (gather_ms > 0 and links > 0) else None},
                "gathered_rows_checked": "checksum of every rank's rows against the solving rank's own checksum" if gather_mode != "none" else None,
            }
            mg = line["multi_gpu"]
            # like-for-like reference for the 1 -> N curve: ONE GPU's rate on this same config = its shard's kernel-only rate
            # (at N = 1 this config has no collective); `--gpus 1` itself runs config 2.  No scaling curve has been measured by
            # the builder (one GPU per lease): these are the figures to read the driver's curve with.
            n1 = units / (kernel_ms * 1e-3)
            mg["n1_same_config"] = {"solves_per_s": n1, "what": f"config {cfg} on one GPU = this run's kernel-only rate per GPU (max over ranks of the kernel time)"}
            # (the kernel-only entry cannot be an efficiency measured here: n1 IS this run's per-GPU kernel rate, so that ratio is 1 by
            # construction; what can be said without a one-GPU run of the same job is how this run's per-GPU kernel rate compares with
            # the one-GPU figure the driver measured for the same config and size last round, a different box and day)
            eff = {"end_to_end": mg["end_to_end_solves_per_s"] / (n1 * world),
                   "kernel_only": None,
                   "kernel_only_note": "not measurable inside an N-GPU run (n1 is taken from it); see per_gpu_kernel_rate_vs_recorded_single_gpu"}
            rec = RECORDED_SINGLE_GPU.get((cfg, n))
            if rec:
                eff["per_gpu_kernel_rate_vs_recorded_single_gpu"] = {"ratio": n1 / rec["solves_per_s"], **rec}
            mg["scaling_efficiency_vs_n1_same_config"] = eff
            if gather_mode != "none" and gather_ms > 0:
                # north star: "all-gather ... only for the final joint array": K sharded steps, ONE all-gather of the last result
                o_ms = other_form["elapsed_s"] * 1e3 if other_form else None
                o_from = (f"a second timed leg of this run: the same K = {args.steps} steps, the same barrier + synchronise bracket, max over ranks"
                          + (f", {other_form['chunks']} stripes per shard" if other_form and other_form["mode"] == "step" else "")) if other_form else None
                if gather_mode == "final":
                    mg["gather_final"] = {"ms_for_K_steps_plus_one_gather": elapsed * 1e3, "solves_per_s": value, "one_all_gather_ms": gather_ms,
                                          "from": "the timed region itself (= value)"}
                    if other_form:
                        mg["gather_step"] = {"ms_per_step": o_ms / args.steps, "solves_per_s": units * world * args.steps / (o_ms * 1e-3), "overlapped": True,
                                             "from": o_from}
                    else:
                        mg["gather_step"] = {"ms_per_step": kernel_ms + gather_ms, "solves_per_s": units * world / ((kernel_ms + gather_ms) * 1e-3),
                                             "overlapped": False, "from": "the kernel-only and gather-only legs of this run, one after the other"}
                else:
                    total_final_ms = o_ms if other_form else kernel_ms * args.steps + gather_ms
                    mg["gather_final"] = {"ms_for_K_steps_plus_one_gather": total_final_ms, "solves_per_s": units * world * args.steps / (total_final_ms * 1e-3),
                                          "one_all_gather_ms": gather_ms, "from": o_from or "the kernel-only and gather-only legs of this run"}
                    mg["gather_step"] = {"ms_per_step": elapsed / args.steps * 1e3, "solves_per_s": value, "overlapped": True,
                                         "from": "the timed region itself (= value)"}
                # the two job shapes side by side, each with its efficiency against one GPU on this config; which of them `value` is, and
                # since when (the advisor's round-5 finding: `value` changed its meaning with the default of --gather)
                mg["value_definition"] = {"value_is": "gather_" + gather_mode,
                                          "default_since_round_5": "gather_final (K sharded steps + ONE all-gather of the final joints + state: the north star's job shape)",
                                          "default_in_rounds_1_to_4": "gather_step (an all-gather inside every step, stripe-pipelined): compare those rounds' `value` with gather_step.solves_per_s",
                                          "changed_in_round": 5}
                for shape in ("gather_final", "gather_step"):
                    mg[shape]["efficiency_vs_n1_same_config"] = mg[shape]["solves_per_s"] / (n1 * world)
            mg["group"] = group_info
            # beside `value`: the north star's own job shape (K sharded steps, ONE all-gather of the final joints) end to end, and the
            # efficiency against one GPU on this same config
            line["scaling_efficiency_vs_n1_same_config"] = eff
            if "gather_final" in mg:
                line["gather_final"] = {"solves_per_s": mg["gather_final"]["solves_per_s"], "ms_for_K_steps_plus_one_gather": mg["gather_final"]["ms_for_K_steps_plus_one_gather"],
                                        "efficiency_vs_n1_same_config": mg["gather_final"]["solves_per_s"] / (n1 * world)}
                line["gather_step"] = {"solves_per_s": mg["gather_step"]["solves_per_s"], "ms_per_step": mg["gather_step"]["ms_per_step"],
                                       "overlapped": mg["gather_step"]["overlapped"],
                                       "efficiency_vs_n1_same_config": mg["gather_step"]["solves_per_s"] / (n1 * world)}
                line["value_definition"] = mg["value_definition"]
        if cfg == 5:
            line["metric"] = "IK control steps/sec (ControlIK continuous, r_arm trajectories)"
            line["unit"] = "steps/s"
            line["roofline"]["kernel_ms"] = kernel_ms / 1000  # per control step (one launch walks the 1000 steps of a pass)
            line["roofline"]["algorithmic_bytes_note"] = (
                "154 B per trajectory-step = 96 (goal matrix) + 58 (joints, reachable, state): the trajectory state stays on chip "
                "between the steps of a pass (SURVEY 8d's persistent-loop figure)")
            line["roofline"]["frac_at_286_bytes_state_round_trip_per_step"] = (
                BYTES_PER_STEP_STATE_ROUND_TRIP * units / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
            line["run_forms_seen"] = dict(forms_seen)
            line["value_is"] = ("K passes back to back, consecutive passes overlapping (the next pass's prepare phase beside this pass's tail); every pass "
                                "resets the trajectory state, re-initialises every trajectory and writes every output row; `steady_state.isolated_pass_ms` "
                                "is one pass on its own, `steady_state.launch_forms_ms.eager` K passes without the overlap"
                                if (pipelined[0] and graph is None) else "K passes back to back without overlap between passes")
        if steady is not None:
            line["steady_state"] = steady

        # ---- what bounds the kernel: counters committed with this build, live clock, cold-HBM run
        build_id = hs.build_id()
        line["library_build"] = build_id
        cnt = committed_counters(cfg, n, build_id)
        if cnt and "bytes" in cnt:
            line["roofline"]["traffic"] = cnt["bytes"]
            line["roofline"]["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) / WRITE_SIZE of this build (profiles/r06/counters.json)"
            if "bytes_per_pass_by_kernel" in cnt:  # config 5: `traffic` is per control step like `achieved`; the pass by kernel:
                line["roofline"]["traffic_per_pass_by_kernel"] = cnt["bytes_per_pass_by_kernel"]
        elif cnt and "stale" in cnt:
            line["roofline"]["traffic_note"] = cnt["stale"]
        if not args.no_extras and not grouped and cfg != 5:
            extras = {}
            # (a) cold HBM: rotate over distinct resident copies of the batch whose combined footprint exceeds the
            # 256 MiB Infinity Cache, so that every launch's inputs and outputs really come from / go to HBM
            footprint = bpp * n
            sets_needed = max(3, int(np.ceil(320 * 2**20 / footprint)) + 1)
            if sets_needed <= 64:
                sets = [main_set] + [make_set(main_set["inputs"][0].clone(), None if main_set["inputs"][1] is None else main_set["inputs"][1].clone())
                                     for _ in range(sets_needed - 1)]
                for s in sets:
                    for f in s["launches"]:
                        f()
                torch.cuda.synchronize()
                kc = max(args.steps, 3 * sets_needed)
                kc -= kc % sets_needed
                e0.record()
                for k in range(kc):
                    for f in sets[k % sets_needed]["launches"]:
                        f()
                e1.record()
                torch.cuda.synchronize()
                cold_ms = e0.elapsed_time(e1) / kc
                extras["cold"] = {"kernel_ms": cold_ms, "sets": sets_needed, "footprint_MB_total": sets_needed * footprint / 1e6, "launches": kc}
                line["roofline"]["kernel_ms_cold"] = cold_ms
                line["roofline"]["achieved_cold"] = bpp * n / (cold_ms * 1e-3) / 1e9
                line["roofline"]["frac_cold"] = line["roofline"]["achieved_cold"] / HBM_PEAK_GBS
                line["roofline"]["cold_note"] = (f"{sets_needed} distinct resident batches ({sets_needed * footprint / 2**20:.0f} MiB of inputs + outputs) "
                                                 "solved round-robin: nothing a launch touches is still in the 256 MiB Infinity Cache")
                del sets
            # (a2) SURVEY 8(d): the vendor's 8 TB/s confirmed by a stream copy on THIS box — 1 GiB read + 1 GiB written per copy (four
            # times the Infinity Cache), device to device, and a read-modify-write of the same size; `frac` is reported against both
            try:
                src_c = torch.empty(1 << 27, dtype=f64, device=dev).normal_()
                dst_c = torch.empty_like(src_c)
                for _ in range(3):
                    dst_c.copy_(src_c)
                torch.cuda.synchronize()
                reps_c = 20
                e0.record()
                for _ in range(reps_c):
                    dst_c.copy_(src_c)
                e1.record()
                torch.cuda.synchronize()
                copy_gbs = 2 * src_c.numel() * 8 / (e0.elapsed_time(e1) / reps_c * 1e-3) / 1e9
                for _ in range(3):  # (the first call of a torch kernel loads it)
                    dst_c.add_(1.0)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(reps_c):
                    dst_c.add_(1.0)
                e1.record()
                torch.cuda.synchronize()
                rmw_gbs = 2 * src_c.numel() * 8 / (e0.elapsed_time(e1) / reps_c * 1e-3) / 1e9
                del src_c, dst_c
                extras["hbm_copy"] = {"copy_GBs": copy_gbs, "read_modify_write_GBs": rmw_gbs, "bytes_per_copy": 2 << 30,
                                      "what": "torch copy_ / add_ of a 1 GiB fp64 tensor, read + written bytes over the HIP-event time of 20 back to back"}
                line["roofline"]["peak_measured_copy"] = max(copy_gbs, rmw_gbs)
                line["roofline"]["frac_of_measured_copy"] = line["roofline"]["achieved"] / max(copy_gbs, rmw_gbs)
            except Exception as e:
                extras["hbm_copy"] = {"error": f"{type(e).__name__}: {e}"}
            # (b) the shader clock the chip holds under this kernel (s_memtime / s_memrealtime in a monitor wave on a side
            # stream, no stamp in the product kernel), after >= 2 s of back-to-back launches, and over a short burst
            try:
                ghz_burst = first_launches_clock(hs, launch_all, max(8, min(args.steps, 64)))
                ghz, sustained_ms = sustained_phase(hs, launch_all, 2.0, 0.5)
                extras["clock"] = {"sustained_ghz": ghz, "sustained_kernel_ms": sustained_ms, "first_launches_ghz": ghz_burst,
                                   "method": "rsik_debug_math op 8: d(s_memtime)/d(s_memrealtime) of 64 monitor waves on a side stream"}
            except Exception as e:
                extras["clock"] = {"error": f"{type(e).__name__}: {e}"}
            # (c) vector-instruction issue: executed instructions per wave (PMC, committed with this build) x waves per
            # launch / kernel time, against the NOMINAL issue peak (2.4 GHz) and against the clock just measured
            if cnt and cnt.get("valu_per_wave"):
                vpw = cnt["valu_per_wave"]
                ach = (n / 64) * vpw / (kernel_ms * 1e-3)
                comp = {"bound": "fp64 VALU issue", "valu_instr_per_wave": vpw, "achieved": ach, "unit": "wave-instr/s",
                        "peak_nominal": NOMINAL_VALU_WAVE_INSTR_PER_S, "frac_nominal": ach / NOMINAL_VALU_WAVE_INSTR_PER_S}
                ck = extras.get("clock", {})
                if ck.get("sustained_ghz") and ck.get("sustained_kernel_ms"):
                    ach_s = (n / 64) * vpw / (ck["sustained_kernel_ms"] * 1e-3)
                    peak_s = 1024 * ck["sustained_ghz"] * 1e9 / 4
                    comp["sustained"] = {"achieved": ach_s, "peak_at_measured_clock": peak_s, "frac_at_measured_clock": ach_s / peak_s}
                if not args.no_valu_calibration:
                    fma = fp64_valu_calibration(hs)
                    comp["fma_only_rate"] = fma
                    comp["ratio_to_fma_only_rate"] = ach / fma
                line["roofline"]["compute"] = comp
            line["extras"] = extras
    yield "timed"
    if rank == 0 and sample is not None:
        line["cpu_baseline"] = cpu_baseline(cfg, sample, args.cpu_seconds, gpu=gpu_rows)
    if rank == 0 and not grouped and not args.no_live_traffic:
        # roofline.traffic from THIS run (behind every timed leg: the children have the GPU to themselves)
        lt = live_traffic(cfg, n, args.lib)
        r = line["roofline"]
        if "bytes" in lt:
            r["traffic"] = lt["bytes"] / (1000 if cfg == 5 else 1)  # (config 5: per control step of all trajectories, like `achieved`)
            r["traffic_source"] = lt["source"]
            r["traffic_over_algorithmic"] = lt["bytes"] / (BYTES_PER_POSE[cfg] * units)
            r["traffic_seconds"] = lt["seconds"]
            r.pop("traffic_note", None)
            if cfg == 5:
                r["traffic_per_pass_by_kernel"] = lt["by_kernel"]
        else:
            r["traffic_live_error"] = lt["error"]
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    return line


if __name__ == "__main__":
    main()
