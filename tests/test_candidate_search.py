"""Host-side check of the candidate set the discrete kernel judges instead of walking the theta grid
(grid_theta_candidates, reachy2_symbolic_ik_amd/csrc/rsik_device.hpp): a NumPy restatement of that selection — the two
grid ends, the outside neighbours of the ends of every arc on which is_elbow_ok fails (utils.py:443-465), and the
brackets of the preferred angle in the two situations where the shortcut (utils.py:357-364) fails although the angle
is free — must pick the same theta as the checker's exhaustive utils.get_best_discrete_theta (utils.py:334-396) on
every pose that is not handed to the exhaustive sweep (`fast_ok`).  The kernel itself is compared with the checker in
tests/test_gpu_parity.py (-m gpu); this test pins the ARGUMENT the kernel relies on, on the CPU."""
import numpy as np
import pytest

from oracle import oracle as O

TWO_PI = 2 * np.pi


def _angle_diff(a, b):
    return ((a - b) + np.pi) % TWO_PI - np.pi


def _is_valid(angle, i0, i1):  # utils.py:468-474
    if i0 % TWO_PI == i1 % TWO_PI:
        return True
    if i0 < i1:
        return i0 <= angle <= i1
    return i0 <= angle or angle <= i1


def _elbow_ok(arm, e):  # utils.py:443-465 (effective predicate)
    es = arm.field("elbow_singularity_position")
    return (e[1] * arm.field("side") < -0.2) and (
        e[2] < (e[0] - es[0]) * arm.field("singularity_limit_coeff") + es[2] - arm.field("singularity_offset"))


def candidate_search(arm, sv, i0, i1, nb, pref):
    """(found, theta, fast_ok) as the kernel's per-lane search decides them; None when the shortcut applies."""
    valid = _is_valid(pref, i0, i1)
    if valid and _elbow_ok(arm, sv.get_elbow_position(pref)):
        return None
    pref_free = (not valid) and abs(pref) > np.pi
    if abs(abs(i0) + abs(i1) - TWO_PI) < 1e-5:
        a, b = np.pi / 2, np.pi / 2 + TWO_PI
        pref_free = pref_free or not valid
    else:
        a, b = i0, (i1 if i0 < i1 else i1 + TWO_PI)
    step = (b - a) / (nb - 1)
    fast_ok = step > 1e-9
    # the elbow circle from three of its points: e(theta) = c2 + ra1 cos(theta) + ra2 sin(theta)
    e0, e1, e2 = sv.get_elbow_position(0.0), sv.get_elbow_position(np.pi / 2), sv.get_elbow_position(np.pi)
    c2 = 0.5 * (e0 + e2)
    ra1, ra2 = e0 - c2, e1 - c2
    side, sc = arm.field("side"), arm.field("singularity_limit_coeff")
    es, so = arm.field("elbow_singularity_position"), arm.field("singularity_offset")
    cons = [(side * ra1[1], side * ra2[1], -0.2 - side * c2[1]),
            (ra1[2] - sc * ra1[0], ra2[2] - sc * ra2[0], (es[2] - so - sc * es[0]) - (c2[2] - sc * c2[0]))]
    last = nb - 1
    eps = max(1e-6, 1e-10 / step) if step > 0 else 1.0
    cand = [0, last]

    def bracket(angle):
        nonlocal fast_ok
        pos = ((angle - a) % TWO_PI) / step
        k0 = int(pos)
        frac = pos - k0
        if frac < eps or frac > 1 - eps:
            fast_ok = False
        return k0

    for (A, B, D) in cons:
        R2 = A * A + B * B
        q = R2 - D * D
        if abs(q) < 1e-6 * R2:
            fast_ok = False
        if q > 0:  # the constraint fails on [phi - alpha, phi + alpha]
            phi, al = np.arctan2(B, A), np.arctan2(np.sqrt(q), D)
            cand.append(min(bracket(phi + al) + 1, last))  # first grid point above the upper end
            cand.append(min(bracket(phi - al), last))      # last grid point below the lower end
    if pref_free:
        k0 = bracket(pref)
        cand += [min(k0, last), min(k0 + 1, last)]
    best_d, best_k = np.inf, None
    for k in cand:
        th = b if k == last else k * step + a
        if _elbow_ok(arm, sv.get_elbow_position(th)):
            d = abs(_angle_diff(th, pref))
            if d < best_d or (d == best_d and k < best_k):
                best_d, best_k = d, k
    theta = None if best_k is None else (b if best_k == last else best_k * step + a)
    return best_k is not None, theta, fast_ok


@pytest.mark.parametrize("dvt_tag,offset", [("std", -1.01), ("dvt", 0.03)])
def test_candidate_set_finds_the_grid_minimum(golden_dir, dvt_tag, offset):
    g = np.load(f"{golden_dir}/g4_control_discrete.npz")
    rng = np.random.default_rng(20261003)
    searched = fallback = 0
    for name in ("r_arm", "l_arm"):
        arm = O.Arm(name, singularity_offset=offset)
        sv = O.Solver(arm)
        M = g[f"{dvt_tag}_{name}_M"]
        eul = O.euler_from_matrix_xyz(M)
        default_pref = -4 * np.pi / 6 if name == "r_arm" else -np.pi + 4 * np.pi / 6
        for i in range(0, len(M), 2):
            ok, itv, _ = sv.is_reachable(M[i, :3, 3], eul[i])
            if not ok:
                continue
            nb = int(rng.choice([2, 3, 5, 10, 20, 33, 64, 200]))
            kind = rng.integers(0, 4)
            # the default, a random angle, the left arm's unwrapped mirror image of one (C:252), the interval's own start
            pref = [default_pref, rng.uniform(-np.pi, np.pi), -np.pi - rng.uniform(-np.pi, np.pi), itv[0] + 1e-3][kind]
            res = candidate_search(arm, sv, itv[0], itv[1], nb, float(pref))
            want = sv.best_discrete_theta(itv, nb, float(pref))
            if res is None:
                assert want == (True, float(pref))
                continue
            searched += 1
            if not res[2]:
                fallback += 1
                continue
            assert res[0] == want[0], (name, i, nb, pref, res, want)
            if want[0]:
                assert abs(res[1] - want[1]) < 1e-12, (name, i, nb, pref, res, want)
    assert searched > 600 and fallback < 0.02 * searched, (searched, fallback)
