#!/usr/bin/env python3
"""Per-wave timeline of solve_kernel from a -DRSIK_TIMELINE_PROBE build (diagnostic only):

    hipcc ... -DRSIK_TIMELINE_PROBE rsik_lib.hip -o build/variants/probe_timeline.so
    python scripts/timeline_probe.py --lib build/variants/probe_timeline.so [n]

Every wave reports four 100 MHz timestamps (start, tables staged + pose loaded, outputs staged in LDS, stores issued)
and its hardware slot.  Prints the launch span, the phase durations, how many waves are resident / computing over
time and the share of SIMD-time with no wave in its arithmetic phase.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def _lib_arg():
    """--lib PATH: the probe build to load instead of the in-tree library (must happen before the package loads it)."""
    if "--lib" in sys.argv:
        k = sys.argv.index("--lib")
        from reachy2_symbolic_ik_amd import _abi

        _abi.use_library(sys.argv[k + 1])
        del sys.argv[k: k + 2]


_lib_arg()
import bench  # noqa: E402
from reachy2_symbolic_ik_amd import SymbolicIK  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
ik = SymbolicIK("r_arm")
P, E = bench.make_config2_poses(min(n, 1 << 20))
if n > len(P):
    reps = (n + len(P) - 1) // len(P)
    P, E = np.tile(P, (reps, 1))[:n], np.tile(E, (reps, 1))[:n]
poses = torch.as_tensor(np.ascontiguousarray(np.concatenate([P.T, E.T], axis=0))).cuda()
for _ in range(5):
    res = ik.solver.solve(poses, arm_uniform=0)
torch.cuda.synchronize()
ppt = int(os.environ.get("RSIK_PROBE_PPT", "1"))  # tiles per workgroup of the probed build: the first tile's rows hold the probe
iv = res["interval"].cpu().numpy().reshape(-1, ppt, 4, 64, 2)[:, 0].reshape(-1, 64, 2)
t0, t1 = iv[:, 0, 0], iv[:, 0, 1]
t2, t3 = iv[:, 1, 0], iv[:, 1, 1]
hw, xcc = iv[:, 2, 0].astype(np.uint64), iv[:, 2, 1].astype(np.uint64)
base = t0.min()
t0, t1, t2, t3 = [(t - base) / 100.0 for t in (t0, t1, t2, t3)]  # microseconds
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
slot = (((xcc & 15) * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
print("waves %d   distinct SIMDs %d   waves/SIMD min %d max %d" % (
    len(t0), len(np.unique(slot)), np.bincount(np.unique(slot, return_inverse=True)[1]).min(),
    np.bincount(np.unique(slot, return_inverse=True)[1]).max()))
print("launch span (first wave start -> last stores issued): %.2f us" % t3.max())


def q(x):
    return "median %.2f  p5 %.2f  p95 %.2f  mean %.2f" % (np.median(x), np.percentile(x, 5), np.percentile(x, 95), x.mean())


print("start -> staged (pose loads + table staging + barrier): " + q(t1 - t0))
print("staged -> outputs in LDS (arithmetic):                  " + q(t2 - t1))
print("outputs in LDS -> stores issued:                        " + q(t3 - t2))
print("wave lifetime:                                          " + q(t3 - t0))
grid = np.arange(0.0, t3.max(), 0.25)
res_w = [(np.sum((t0 <= g) & (t3 > g)), np.sum((t1 <= g) & (t2 > g))) for g in grid]
print("time us : resident waves / waves in arithmetic phase (of %d SIMDs)" % len(np.unique(slot)))
for g, (a, b) in list(zip(grid, res_w))[:: max(1, len(grid) // 48)]:
    print("  %6.2f : %6d / %6d" % (g, a, b))
# per-SIMD: fraction of the launch span in which no resident wave is in its arithmetic phase
idle = []
span = t3.max()
for s_ in np.unique(slot):
    m = slot == s_
    ev = sorted([(a, 1) for a in t1[m]] + [(b, -1) for b in t2[m]])
    busy, depth, last = 0.0, 0, 0.0
    for t, d in ev:
        if depth > 0:
            busy += t - last
        depth += d
        last = t
    idle.append(1.0 - busy / span)
idle = np.array(idle)
print("SIMD-time with no wave in its arithmetic phase: mean %.1f %%  (p5 %.1f, p95 %.1f)" % (
    100 * idle.mean(), 100 * np.percentile(idle, 5), 100 * np.percentile(idle, 95)))
starts = np.sort(t0)
print("wave start times, every 1/16 quantile (us):", " ".join("%.1f" % starts[int(k * (len(starts) - 1) / 16)] for k in range(17)))

# per-XCD view: the hardware dispatcher deals workgroups to the 8 XCDs round-robin, so each XCD owns n/8 of the launch
x = (xcc & 15).astype(int)
print("per XCD: waves, distinct SIMDs, last store issued (us), mean arithmetic phase (us), mean start->staged (us)")
for k in np.unique(x):
    m = x == k
    print("  xcd %d: %6d %5d   %6.2f   %5.2f   %5.2f" % (k, m.sum(), len(np.unique(slot[m])), t3[m].max(), (t2 - t1)[m].mean(), (t1 - t0)[m].mean()))
cuid = slot // 4
fin = np.array([t3[cuid == c].max() for c in np.unique(cuid)])
print("per-CU finish time: min %.2f  p25 %.2f  median %.2f  p75 %.2f  max %.2f us" % (
    fin.min(), np.percentile(fin, 25), np.median(fin), np.percentile(fin, 75), fin.max()))
out = os.environ.get("RSIK_TIMELINE_NPZ")
if out:
    np.savez_compressed(out, t0=t0, t1=t1, t2=t2, t3=t3, hw=hw, xcc=xcc)
