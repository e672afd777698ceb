#!/usr/bin/env python3
"""Diagnostic: the constructor's default start (arm along the body = fully extended, degenerate elbow circle) through
the scalar drop-in API on the GPU: circle radius, the start-up ternary search's distances step by step."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from reachy2_symbolic_ik_amd import SymbolicIK  # noqa: E402

np.set_printoptions(precision=17)


def adiff(a, b):
    return ((a - b + np.pi) % (2 * np.pi)) - np.pi


for arm, cj, y in (("r_arm", [0.0, 0.2617993877991494, -0.17453292519943295, 0, 0, 0, 0], -0.2),):
    s = bench._quiet(SymbolicIK, arm=arm, singularity_offset=-1.01)
    pose = np.array([[0, y, -0.66], [0, 0, 0]])
    ok, itv, f = s.is_reachable_no_limits(pose)
    print(arm, "circle centre", s.intersection_circle[0], "radius", repr(s.intersection_circle[1]), "normal", s.intersection_circle[2])
    print("wrist", s.wrist_position, "goal", s.goal_pose)
    pref = -4 * np.pi / 6
    low, high = -np.pi, np.pi

    def dist(j):
        return np.linalg.norm([adiff(j[i], cj[i]) for i in range(7)])

    j, _ = f(pref)
    print("pref dist", repr(dist(j)), j)
    it = 0
    while (high - low) > 0.01:
        mid1 = low + (high - low) / 3
        mid2 = high - (high - low) / 3
        j1, _ = f(mid1)
        j2, _ = f(mid2)
        f1, f2 = dist(j1), dist(j2)
        print(it, repr(low), repr(high), "f1-f2 = %.3e" % (f1 - f2), repr(f1), repr(f2))
        if it in (0, 14, 15):
            print("   j1", j1, "\n   j2", j2)
        if f1 < f2:
            high = mid2
        else:
            low = mid1
        it += 1
    print("best", (low + high) / 2)
