"""Host-side check of the argument the theta phase of the trajectory pipeline relies on (continuous_next_theta_lean,
reachy2_symbolic_ik_amd/csrc/rsik_device.hpp; theta_snap_plan, rsik_lib.hip): limit_theta_to_interval's choice of the
nearer interval end — abs(angle_diff(theta, l1)) < abs(angle_diff(theta, l0)), utils.py:105-111 — is, for a theta in
(-pi, pi] outside the interval, ONE comparison with a threshold, and the whole snap is min / max arithmetic around it:

    inner interval (l0 < l1):       theta < t ? min(max(theta, l0), l1) : l0
    wrap-around interval (l0 > l1): theta < t ? min(theta, l1) : max(theta, l0)

with t found by bisection over the doubles of the gap.  Checked here with Python's own float arithmetic (the reference's)
for the four control intervals of ControlIK (control_ik.py:252-266, both arms x both modes) and for random ones; the
kernel itself is compared bit for bit with the reference's sequence of operations in tests/test_gpu_parity.py (-m gpu)."""
import math

import numpy as np
import pytest

PI = math.pi
TWO_PI = 2 * math.pi


def angle_diff(a, b):  # utils.py:486-490
    return ((a - b) + PI) % TWO_PI - PI


def is_valid_angle(angle, i0, i1):  # utils.py:468-474
    if i0 % TWO_PI == i1 % TWO_PI:
        return True
    if i0 < i1:
        return i0 <= angle <= i1
    return i0 <= angle or angle <= i1


def limit_after_wrap(theta, l0, l1):  # utils.py:98-112 (theta already wrapped to (-pi, pi])
    if is_valid_angle(theta, l0, l1):
        return theta
    return l1 if abs(angle_diff(theta, l1)) < abs(angle_diff(theta, l0)) else l0


def snap_plan(l0, l1):
    """(kind, t) as theta_snap_plan derives them: kind 'inner' / 'wrap', or None where the kernel keeps the generic step."""
    if not (abs(l0) <= PI and abs(l1) <= PI) or l0 == l1 or (abs(l0) == PI and abs(l1) == PI):
        return None, 0.0

    def nearer_is_l1(t):
        return abs(angle_diff(t, l1)) < abs(angle_diff(t, l0))

    wrap = not (l0 < l1)
    lo, hi = l1, (l0 if wrap else PI)
    if not lo < hi:
        return None, 0.0
    lo = math.nextafter(lo, hi)
    if not nearer_is_l1(lo) or nearer_is_l1(hi):
        return None, 0.0
    while math.nextafter(lo, hi) < hi:
        mid = lo + (hi - lo) / 2
        if nearer_is_l1(mid):
            lo = mid
        else:
            hi = mid
    return ("wrap" if wrap else "inner"), hi


def snapped(theta, l0, l1, kind, t):
    if kind == "inner":
        return min(max(theta, l0), l1) if theta < t else l0
    return min(theta, l1) if theta < t else max(theta, l0)


def control_limits(arm, mode):  # control_ik.py:252-266 as rsik_lib.hip's control_limits restates it
    l0, l1 = (3 * PI / 4, -2 * PI / 6) if mode == "unconstrained" else (-4 * PI / 5, 0.0)
    if arm == "l_arm":
        l0, l1 = -PI - l1, -PI - l0
        l0 = l0 % TWO_PI if l0 < -PI else l0
        l1 = l1 % TWO_PI if l1 < -PI else l1
        l0 = l0 % -TWO_PI if l0 > PI else l0
        l1 = l1 % -TWO_PI if l1 > PI else l1
    return l0, l1


def check_interval(l0, l1, rng, samples=20000):
    kind, t = snap_plan(l0, l1)
    if kind is None:
        return None
    pts = list(rng.uniform(-PI, PI, samples)) + [PI, l0, l1, t]
    for base in (l0, l1, t, PI, -PI):
        x = base
        for _ in range(64):
            x = math.nextafter(x, -4.0)
            pts.append(x)
        x = base
        for _ in range(64):
            x = math.nextafter(x, 4.0)
            pts.append(x)
    for theta in pts:
        if not -PI < theta <= PI:
            continue
        want = limit_after_wrap(theta, l0, l1)
        got = snapped(theta, l0, l1, kind, t)
        assert want == got, (l0, l1, kind, t, theta, want, got)
    return kind


@pytest.mark.parametrize("arm", ["r_arm", "l_arm"])
@pytest.mark.parametrize("mode", ["unconstrained", "low_elbow"])
def test_control_intervals_snap_with_one_threshold(arm, mode):
    l0, l1 = control_limits(arm, mode)
    kind = check_interval(l0, l1, np.random.default_rng(len(arm) * 7 + len(mode)))
    assert kind == ("wrap" if (arm, mode) == ("r_arm", "unconstrained") else "inner")


def test_random_intervals_snap_with_one_threshold_or_are_refused():
    rng = np.random.default_rng(20261004)
    kinds = {"inner": 0, "wrap": 0, None: 0}
    for _ in range(60):
        l0, l1 = rng.uniform(-PI, PI, 2)
        kinds[check_interval(float(l0), float(l1), rng, samples=3000)] += 1
    assert kinds["inner"] > 5 and kinds["wrap"] > 5, kinds  # (refused: the gap's nearer-end boundary lies below l0)
