#!/usr/bin/env python3
"""Per-wave timeline of control_discrete_kernel from a -DRSIK_TIMELINE_PROBE build (diagnostic only):

    python scripts/build_variant.py probe_timeline -DRSIK_TIMELINE_PROBE
    python scripts/disc_timeline_probe.py --lib build/variants/probe_timeline.so [n]

Every wave reports six 100 MHz timestamps — start, inputs + tables in (head), reach + preferred-theta shortcut done,
theta chosen (grid search), joints + safety_checks done, stores acknowledged — and its hardware slot.  Prints the launch
span, the phase durations, how many waves are in which phase over time, and what the launch would cost if every SIMD
only ever issued (the phase sums per SIMD)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:
    k = sys.argv.index("--lib")
    from reachy2_symbolic_ik_amd import _abi

    _abi.use_library(sys.argv[k + 1])
    del sys.argv[k: k + 2]
import bench  # noqa: E402
from reachy2_symbolic_ik_amd import ControlIK  # noqa: E402
from reachy2_symbolic_ik_amd.control_ik import matrices_to_m12_soa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
M = bench.make_config3_matrices(min(n, 1 << 18))
if n > len(M):
    M = np.tile(M, ((n + len(M) - 1) // len(M), 1, 1))[:n]
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF)
ctrl.nb_search_points = 64
m12 = matrices_to_m12_soa(M, torch.device("cuda", 0))
for _ in range(5):
    res = ctrl.symbolic_inverse_kinematics_batch("r_arm", m12)
torch.cuda.synchronize()
J = res["joints"].cpu().numpy().reshape(-1, 64, 7)
T = J[:, 0, :6]
hw, xcc = J[:, 0, 6].astype(np.uint64), J[:, 1, 0].astype(np.uint64)
base = T[:, 0].min()
T = (T - base) / 100.0  # microseconds
names = ["head: 12 columns + tables in", "reach + shortcut", "grid search", "joints + safety", "stores acknowledged"]
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
slot = (((xcc & 15) * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
uniq, inv = np.unique(slot, return_inverse=True)
per = np.bincount(inv)
print(f"n = {n}: {len(T)} waves on {len(uniq)} SIMDs (waves per SIMD min {per.min()} max {per.max()})")
print(f"launch span (first wave start -> last stores acknowledged): {T[:, 5].max():.2f} us; last wave starts at {T[:, 0].max():.2f} us")
d = np.diff(T, axis=1)
for k, nm in enumerate(names):
    print(f"  {nm:32s} mean {d[:, k].mean():6.2f} us   median {np.median(d[:, k]):6.2f}   p90 {np.percentile(d[:, k], 90):6.2f}   max {d[:, k].max():6.2f}")
print(f"  wave lifetime                    mean {(T[:, 5] - T[:, 0]).mean():6.2f} us")
grid = np.arange(0.0, T[:, 5].max() + 0.25, 0.25)
print("time_us  resident  in_head  computing(reach..safety)  storing")
for g in grid[::4]:
    res_ = ((T[:, 0] <= g) & (T[:, 5] > g)).sum()
    head = ((T[:, 0] <= g) & (T[:, 1] > g)).sum()
    comp = ((T[:, 1] <= g) & (T[:, 4] > g)).sum()
    sto = ((T[:, 4] <= g) & (T[:, 5] > g)).sum()
    print(f"{g:7.2f}  {res_:8d}  {head:7d}  {comp:9d}  {sto:16d}")
# per-SIMD: time with at least one wave in an arithmetic phase vs the launch span
busy = np.zeros(len(uniq))
for s_ in range(len(uniq)):
    iv = sorted((a, b) for a, b in zip(T[inv == s_, 1], T[inv == s_, 4]))
    end, tot = -1.0, 0.0
    for a, b in iv:
        if a > end:
            tot += b - a
            end = b
        elif b > end:
            tot += b - end
            end = b
    busy[s_] = tot
print(f"per SIMD: time with >= 1 wave in reach..safety: mean {busy.mean():.2f} us of the {T[:, 5].max():.2f} us span "
      f"({100 * busy.mean() / T[:, 5].max():.0f} %); sum of the arithmetic phases of its waves: mean {np.bincount(inv, weights=(T[:, 4] - T[:, 1])).mean():.2f} us")
# how uneven the single-round launch is: when each SIMD / CU / XCD finishes its last wave
fin = np.array([T[inv == s_, 5].max() for s_ in range(len(uniq))])
print("SIMD finish time (us): min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f" % (fin.min(), np.percentile(fin, 10), np.median(fin), np.percentile(fin, 90), fin.max()))
cu_id = uniq // 4
cfin = np.array([fin[cu_id == c].max() for c in np.unique(cu_id)])
print("CU finish time   (us): min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f" % (cfin.min(), np.percentile(cfin, 10), np.median(cfin), np.percentile(cfin, 90), cfin.max()))
xcd = (uniq // (8 * 2 * 16 * 4))
print("XCD finish time  (us):", " ".join("%.2f" % fin[xcd == x].max() for x in np.unique(xcd)))
work = (T[:, 4] - T[:, 1])
print("per-wave arithmetic phase (us, SIMD shared 4 ways): p10 %.2f median %.2f p90 %.2f max %.2f" % tuple(np.percentile(work, [10, 50, 90, 100])))
start = np.array([T[inv == s_, 1].min() for s_ in range(len(uniq))])
print("SIMD: first wave computing at (us): p10 %.2f median %.2f p90 %.2f; busy span (first compute -> last store): median %.2f p90 %.2f" % (
    np.percentile(start, 10), np.median(start), np.percentile(start, 90), np.median(fin - start), np.percentile(fin - start, 90)))
# lanes that went through the grid search, per wave, against the wave's finish time
wx = (slot // (8 * 2 * 16 * 4))
print("per XCD: first wave start / median wave start / median 'inputs in' / last finish (us):")
for x in np.unique(wx):
    m = wx == x
    print("   XCD %d: %5.2f %5.2f %5.2f %6.2f   waves %d" % (x, T[m, 0].min(), np.median(T[m, 0]), np.median(T[m, 1]), T[m, 5].max(), m.sum()))
