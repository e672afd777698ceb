#!/usr/bin/env python3
"""Per-stage device time of the two batch kernels (SURVEY f-4: the batched counterpart of the per-function timers of the reference's
src/benchmark/ik_benchmarks.py:12-156), as one JSON object on stdout — what `bench.py --stages` embeds and profiles/r04/stage_timers.json holds.

  config 2 (rsik_solve, 1 Mi reachable poses)      config 3 (rsik_control_discrete, 256 Ki goal matrices, 64-point sweep)
    head      pose loads + tables staged               head      twelve columns + tables in
    goal      euler -> the goal's three vectors        reach     goal vectors + is_reachable + the preferred-theta shortcut
    reach     is_reachable: in-reach test, wrist,      search    the theta grid search (utils.py:366-396)
              the two circles, their interval          joints    get_joints(theta) (symbolic_ik.py:697-863)
    joints    get_joints(interval[0])                  safety    limit_orbita3d_joints + multiturn checks (utils.py:443-474, 535-568)
    stores    rows out                                 stores    rows out, acknowledged

Two kinds of figures:
  launches   the PRODUCT library, HIP events around K launches: rsik_solve with theta policy "none" (is_reachable alone: reach only /
             + its interval row written) against the full solve — stage cost as a difference of whole launches;
  waves      a -DRSIK_TIMELINE_PROBE build (diagnostic: s_memrealtime stamps at the stage boundaries, 100 MHz): per wave, the time
             between boundaries — a wave shares its SIMD with up to four others, so a stage's time is its share of the wave's
             lifetime, not its issue time; the instruction counts per stage (profiles/r04/isa_sections_*.txt) give that.
usage: stage_timers.py [--probe-lib build/variants/probe_timeline.so] [--no-build]   (builds the probe variant if it is missing)"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROBE = os.path.join(ROOT, "build", "variants", "probe_timeline.so")

CHILD = r'''
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, %(root)r)
from reachy2_symbolic_ik_amd import _abi
_abi.use_library(sys.argv[1])
import bench
from reachy2_symbolic_ik_amd import ControlIK, SymbolicIK
from reachy2_symbolic_ik_amd.control_ik import matrices_to_m12_soa
out = {}
def stats(d):  # d [waves, stages] in us
    return {"mean_us": [float(v) for v in d.mean(axis=0)], "median_us": [float(v) for v in np.median(d, axis=0)]}
# ---- config 2
ik = bench._quiet(SymbolicIK, "r_arm")
P, E = bench.make_config2_poses(1 << 20)
poses = torch.as_tensor(np.ascontiguousarray(np.concatenate([P.T, E.T], axis=0))).cuda()
for _ in range(5):
    res = ik.solver.solve(poses, arm_uniform=0)
torch.cuda.synchronize()
iv = res["interval"].cpu().numpy().reshape(-1, 4, 64, 2).reshape(-1, 64, 2)
T = np.stack([iv[:, 0, 0], iv[:, 0, 1], iv[:, 3, 0], iv[:, 3, 1], iv[:, 1, 0], iv[:, 1, 1]], axis=1)  # start, head, goal, reach, joints, stores
span = (T[:, 5].max() - T[:, 0].min()) / 100.0
d = np.diff(T, axis=1) / 100.0
out["config2"] = {"stages": ["head", "goal", "reach", "joints", "stores"], "waves": int(len(T)), "launch_span_us": float(span),
                  "wave_lifetime_mean_us": float(((T[:, 5] - T[:, 0]) / 100.0).mean()), **stats(d)}
# ---- config 3
M = bench.make_config3_matrices(1 << 18)
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF)
ctrl.nb_search_points = 64
m12 = matrices_to_m12_soa(M, torch.device("cuda", 0))
for _ in range(5):
    res = ctrl.symbolic_inverse_kinematics_batch("r_arm", m12)
torch.cuda.synchronize()
J = res["joints"].cpu().numpy().reshape(-1, 64, 7)
t = J[:, 0, :6]
T = np.stack([t[:, 0], t[:, 1], t[:, 2], t[:, 3], J[:, 1, 1], t[:, 4], t[:, 5]], axis=1)  # start, head, reach, search, joints, safety, stores
span = (T[:, 6].max() - T[:, 0].min()) / 100.0
d = np.diff(T, axis=1) / 100.0
out["config3"] = {"stages": ["head", "reach", "search", "joints", "safety", "stores"], "waves": int(len(T)), "launch_span_us": float(span),
                  "wave_lifetime_mean_us": float(((T[:, 6] - T[:, 0]) / 100.0).mean()), **stats(d)}
print("STAGES " + json.dumps(out))
'''


def launches(k=200):
    """The product library: is_reachable alone (with / without its interval row) against the full solve, K launches each."""
    import torch

    import bench
    from reachy2_symbolic_ik_amd import SymbolicIK, _abi

    ik = bench._quiet(SymbolicIK, "r_arm")
    hs = ik.solver
    n = 1 << 20
    P, E = bench.make_config2_poses(n)
    poses = torch.as_tensor(np.ascontiguousarray(np.concatenate([P.T, E.T], axis=0))).cuda()
    f64, u8 = torch.float64, torch.uint8
    o = {"joints": torch.empty((n, 7), dtype=f64, device="cuda"), "interval": torch.empty((n, 2), dtype=f64, device="cuda"),
         "reachable": torch.empty((n,), dtype=u8, device="cuda"), "state": torch.empty((n,), dtype=u8, device="cuda")}
    cols = [poses[k_] for k_ in range(6)]
    import ctypes as C

    cptr = (C.c_void_p * 6)(*[c.data_ptr() for c in cols])

    def call(policy, joints, interval):
        hs._check(hs.lib.rsik_solve(hs._h, n, cptr, None, 0, policy, None, None, joints, interval, None, o["reachable"].data_ptr(), o["state"].data_ptr()))

    forms = {"reach_only": (_abi.THETA_NONE, None, None), "reach_and_interval": (_abi.THETA_NONE, None, o["interval"].data_ptr()),
             "full_solve": (_abi.THETA_INTERVAL0, o["joints"].data_ptr(), o["interval"].data_ptr())}
    res = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.device(hs.device):
        hs._bind_stream()
        for _ in range(3):  # (three rounds, the best of each form: the clock is power-managed)
            for name, (pol, j, iv) in forms.items():
                for _ in range(10):
                    call(pol, j, iv)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(k):
                    call(pol, j, iv)
                e1.record()
                torch.cuda.synchronize()
                res[name] = min(res.get(name, 1e9), e0.elapsed_time(e1) / k * 1e3)
    return {"poses": n, "launches_each": k, "kernel_us": res,
            "joints_stage_us_by_difference": res["full_solve"] - res["reach_and_interval"],
            "note": "HIP events around K back-to-back launches of the product library; a stage = the difference of two whole launches"}


def main():
    probe = PROBE
    if "--probe-lib" in sys.argv:
        probe = os.path.abspath(sys.argv[sys.argv.index("--probe-lib") + 1])
    def stale(path):  # built from another tree?  the library carries the hash of its sources (asked in a child: this process loads
        # the product library, and two builds of one library in a process is a thing to avoid)
        from reachy2_symbolic_ik_amd import build as B

        code = "import ctypes as C,sys; l=C.CDLL(sys.argv[1]); l.rsik_build_id.restype=C.c_char_p; print(l.rsik_build_id().decode())"
        p = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True)
        return p.returncode != 0 or B.source_hash() not in p.stdout

    if (not os.path.exists(probe) or stale(probe)) and "--no-build" not in sys.argv:
        subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "build_variant.py"), "probe_timeline", "-DRSIK_TIMELINE_PROBE"],
                       check=True, stdout=subprocess.DEVNULL)
    out = {"launches_config2": launches()}
    if os.path.exists(probe) and stale(probe):
        out["waves"] = {"error": "the probe build is not of this tree's sources (rebuild: scripts/build_variant.py probe_timeline -DRSIK_TIMELINE_PROBE)"}
    elif os.path.exists(probe):
        p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, probe], capture_output=True, text=True, timeout=600)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("STAGES ")]
        if p.returncode != 0 or not line:
            out["waves"] = {"error": (p.stderr or p.stdout)[-800:]}
        else:
            out["waves"] = json.loads(line[0][7:])
            out["waves"]["note"] = ("-DRSIK_TIMELINE_PROBE build: s_memrealtime (100 MHz) at the stage boundaries of every wave; a stage's figure is the "
                                    "time the wave spent between two boundaries while sharing its SIMD with the other resident waves")
    else:
        out["waves"] = {"error": "no probe build (build/variants/probe_timeline.so) and --no-build"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
