#!/usr/bin/env python3
"""Throughput of back-to-back INDEPENDENT 1 M-pose batches issued on one stream vs alternately on two:
with two chains in flight the head of one launch (waves waiting for their first loads) and the tail of another (SIMDs
running out of waves) are filled by the other chain's steady state.  Not what bench.py reports: there the K steps run
one after the other and `roofline` quotes a kernel's own duration.

    python scripts/two_streams.py [steps] [config: 2 (default) | 3 = 262 144 goal matrices through ControlIK discrete, 64-point grid]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from reachy2_symbolic_ik_amd import SymbolicIK  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
CFG = int(sys.argv[2]) if len(sys.argv) > 2 else 2
plans = []
if CFG == 2:
    n = 1 << 20
    ik = bench._quiet(SymbolicIK, "r_arm")
    P, E = bench.make_config2_poses(n)
    dev = ik.solver.device
    hs = ik.solver
    for _ in range(2):
        soa = torch.as_tensor(np.ascontiguousarray(np.concatenate([P.T, E.T], axis=0))).to(dev)
        out = {"joints": torch.empty((n, 7), dtype=torch.float64, device=dev), "interval": torch.empty((n, 2), dtype=torch.float64, device=dev),
               "reachable": torch.empty((n,), dtype=torch.uint8, device=dev), "state": torch.empty((n,), dtype=torch.uint8, device=dev)}
        plans.append(ik.solve_batch(soa, want_elbow=False, out=out, plan_only=True))
else:
    from reachy2_symbolic_ik_amd import ControlIK
    from reachy2_symbolic_ik_amd.control_ik import matrices_to_m12_soa

    n = 1 << 18
    ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
    ctrl.nb_search_points = 64
    hs = ctrl._solver
    dev = hs.device
    M = bench.make_config3_matrices(n)
    for _ in range(2):
        m12 = matrices_to_m12_soa(M, dev).clone()
        out = {"joints": torch.empty((n, 7), dtype=torch.float64, device=dev), "reachable": torch.empty((n,), dtype=torch.uint8, device=dev),
               "state": torch.empty((n,), dtype=torch.uint8, device=dev), "emergency": torch.empty((n,), dtype=torch.uint8, device=dev)}
        plans.append(ctrl.symbolic_inverse_kinematics_batch("r_arm", m12, out=out, plan_only=True))


def capture(n_streams):
    side = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        main = torch.cuda.current_stream(dev)
        for s in side:
            s.wait_stream(main)
        for k in range(K):
            plans[k % 2]["launch"](side[k % n_streams].cuda_stream)  # (plans are stream-bound: name the capture's stream)
        for s in side:
            main.wait_stream(s)
    hs._bind_stream()
    return g


for n_streams in (1, 2):
    g = capture(n_streams)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("%d stream(s): %d batches of %d poses in %.2f ms = %.2f us per batch, %.2f G solves/s" % (
        n_streams, K, n, ms, ms / K * 1e3, K * n / (ms * 1e-3) / 1e9))
if CFG == 2:
    assert int(plans[0]["reachable"].sum()) == n and int(plans[1]["reachable"].sum()) == n
else:
    assert torch.equal(plans[0]["state"], plans[1]["state"]) and int(plans[0]["reachable"].sum()) > 0
