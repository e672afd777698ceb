#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE — runs only in the build container).

Imports the *real* reference (pollen-robotics/reachy2_symbolic_ik, mounted read-only
at /root/reference) and records inputs + expected outputs of the analytic solve path
as small .npz fixtures under tests/golden/.  Nothing of the reference's source is
written to the repo: only numbers.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [--out tests/golden]

Environment used for the committed fixtures: Python 3.10.12, numpy 2.2.6,
scipy 1.15.3 (the reference pins scipy==1.8.0, not installable offline; see
DESIGN.md "oracle pinning").

Reference entry points exercised (file:line in /root/reference/src/reachy2_symbolic_ik):
  SymbolicIK.__init__                 symbolic_ik.py:26-83
  SymbolicIK.is_reachable             symbolic_ik.py:121-282
  SymbolicIK.is_reachable_no_limits   symbolic_ik.py:85-119
  SymbolicIK.get_joints               symbolic_ik.py:697-863
  SymbolicIK.get_elbow_position       symbolic_ik.py:684-695
  ControlIK.__init__                  control_ik.py:28-160
  ControlIK.symbolic_inverse_kinematics (discrete, continuous)  control_ik.py:162-497

Canonical semantics: every recorded get_joints() is preceded by a FRESH
is_reachable() on the same solver object.
"""
import argparse
import contextlib
import io
import os
import sys
import time as _time

import numpy as np

REF_SRC = "/root/reference/src"
sys.dont_write_bytecode = True
sys.path.insert(0, REF_SRC)

from scipy.spatial.transform import Rotation as R  # noqa: E402

import reachy2_symbolic_ik.control_ik as ref_control_mod  # noqa: E402
from reachy2_symbolic_ik.control_ik import ControlIK  # noqa: E402
from reachy2_symbolic_ik.symbolic_ik import SymbolicIK  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import scale_inputs as SCALE  # noqa: E402  (seeded input generators of G14: NumPy + libm only)

# state string <-> uint8 code (shared with include/rsik.h, keep in sync)
STATE_CODES = {
    "reachable": 0,
    "Pose out of reach": 1,
    "Backward pose": 2,
    "wrist out of range": 3,
    "limited by wrist": 4,
    "out of reach - should not happen": 5,
    "limited by shoulder": 6,
    "": 7,
}

ARMS = ["r_arm", "l_arm"]
SHOULDER = {"r_arm": np.array([0.0, -0.2, 0.0]), "l_arm": np.array([0.0, 0.2, 0.0])}
NAN = float("nan")


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def make_solver(arm, singularity_offset=0.03):
    return quiet(SymbolicIK, arm=arm, singularity_offset=singularity_offset)


def mirror_pose(pos, eul):
    """r <-> l mirror rule (src/example/test_random_reachability.py:156-166)."""
    return np.array([pos[0], -pos[1], pos[2]]), np.array([-eul[0], eul[1], -eul[2]])


def solve_symbolic(solver, pos, eul, theta=None, theta_u=None, prev=None):
    """One canonical solve: fresh is_reachable, then ONE get_joints.

    theta policy: explicit theta, or interval[0] (theta_u None), or a point inside the
    interval at fraction theta_u in [0,1).
    Returns dict of plain numbers (NaN where the reference returns [] / None).
    """
    pose = np.array([np.array(pos, dtype=float), np.array(eul, dtype=float)])
    ok, interval, fn, state = solver.is_reachable(pose)
    out = {
        "reachable": np.uint8(bool(ok)),
        "state": np.uint8(STATE_CODES[state]),
        "interval": np.array([NAN, NAN]),
        "theta": NAN,
        "joints": np.full(7, NAN),
        "elbow": np.full(3, NAN),
        "elbow_len": np.uint8(0),
    }
    if ok:
        out["interval"] = np.array(interval, dtype=float)
        if theta is None:
            if theta_u is None:
                theta = float(interval[0])
            else:
                a, b = float(interval[0]), float(interval[1])
                if a > b:
                    b += 2 * np.pi
                theta = a + theta_u * (b - a)
        out["theta"] = theta
        if prev is None:
            joints, elbow = fn(theta)
        else:
            joints, elbow = fn(theta, list(prev))
        out["joints"] = np.array(joints, dtype=float)
        out["elbow"] = np.array(elbow[:3], dtype=float)
        out["elbow_len"] = np.uint8(len(elbow))  # Q2: 4 normally, 3 if projection fired
    return out


def stack(dicts):
    keys = dicts[0].keys()
    return {k: np.stack([np.asarray(d[k]) for d in dicts]) for k in keys}


def random_poses(rng, arm, n):
    pos = SHOULDER[arm] + rng.uniform(-0.7, 0.7, size=(n, 3))
    eul = rng.uniform(-np.pi, np.pi, size=(n, 3))
    return pos, eul


def reachable_poses(rng, solver, arm, n):
    """Rejection-sample poses whose is_reachable state is 'reachable'."""
    P, E = [], []
    while len(P) < n:
        pos, eul = random_poses(rng, arm, 4096)
        for p, e in zip(pos, eul):
            ok, _, _, _ = solver.is_reachable(np.array([p, e]))
            if ok:
                P.append(p)
                E.append(e)
                if len(P) == n:
                    break
    return np.array(P), np.array(E)


def pose_to_matrix(pos, eul):
    M = np.eye(4)
    M[:3, :3] = R.from_euler("xyz", eul).as_matrix()
    M[:3, 3] = pos
    return M


# ----------------------------------------------------------------------------------------
# G0: per-arm constants
# ----------------------------------------------------------------------------------------
def gen_constants(out):
    data = {}
    for arm in ARMS:
        for tag, so in (("dflt", 0.03), ("ctrl", -1.01)):
            s = make_solver(arm, so)
            p = f"{arm}_{tag}_"
            data[p + "shoulder_position"] = np.array(s.shoulder_position, dtype=float)
            data[p + "shoulder_orientation_offset"] = np.array(s.shoulder_orientation_offset, dtype=float)
            data[p + "upper_arm_size"] = np.float64(s.upper_arm_size)
            data[p + "forearm_size"] = np.float64(s.forearm_size)
            data[p + "tip_position"] = np.array(s.tip_position, dtype=float)
            data[p + "gripper_size"] = np.float64(s.gripper_size)
            data[p + "max_arm_length"] = np.float64(s.max_arm_length)
            data[p + "shoulder_wrist_min_distance"] = np.float64(s.shoulder_wrist_min_distance)
            data[p + "elbow_singularity_position"] = np.array(s.elbow_singularity_position, dtype=float)
            data[p + "wrist_singularity_position"] = np.array(s.wrist_singularity_position, dtype=float)
            data[p + "singularity_offset"] = np.float64(s.singularity_offset)
            data[p + "singularity_limit_coeff"] = np.float64(s.singularity_limit_coeff)
            data[p + "wrist_limit"] = np.float64(s.wrist_limit)
            data[p + "backward_limit"] = np.float64(s.backward_limit)
            data[p + "projection_margin"] = np.float64(s.projection_margin)
            data[p + "normal_vector_margin"] = np.float64(s.normal_vector_margin)
            data[p + "elbow_limit"] = np.float64(s.elbow_limit)
    # URDF-derived parameters (utils.py:661-690) for the host-side parser test
    c = quiet(ControlIK, urdf_path="../config_files/reachy2.urdf")
    for arm in ARMS:
        s = c.symbolic_ik_solver[arm]
        p = f"{arm}_urdf_"
        data[p + "shoulder_position"] = np.array(s.shoulder_position, dtype=float)
        data[p + "shoulder_orientation_offset"] = np.array(s.shoulder_orientation_offset, dtype=float)
        data[p + "upper_arm_size"] = np.float64(s.upper_arm_size)
        data[p + "forearm_size"] = np.float64(s.forearm_size)
        data[p + "tip_position"] = np.array(s.tip_position, dtype=float)
        data[p + "elbow_singularity_position"] = np.array(s.elbow_singularity_position, dtype=float)
        data[p + "wrist_limit"] = np.float64(s.wrist_limit)
        data[p + "previous_theta_init"] = np.float64(c.previous_theta[arm])
        data[p + "preferred_theta"] = np.float64(c.preferred_theta[arm])
        data[p + "previous_sol"] = np.array(c.previous_sol[arm], dtype=float)
    np.savez_compressed(os.path.join(out, "g0_constants.npz"), **data)


# ----------------------------------------------------------------------------------------
# G1: known-answer catalogue
# ----------------------------------------------------------------------------------------
def catalogue():
    d2r = np.radians
    cat = []
    # tests/test_ik.py:17-79 (r_arm, default solver) — the reference's own unit-test poses
    t_ik = [
        ([0.4, 0.2, 0.1], [d2r(-60), d2r(-90), d2r(20)]),
        ([0.3, -0.2, -0.3], [d2r(0), d2r(-90), d2r(0)]),
        ([0.02, -0.2, -0.65], [0.0, 0.0, 0.0]),
        ([0.0, -0.2, -0.65], [0.0, 0.0, 0.0]),
        ([0.87, -0.2, -0.0], [0.0, -np.pi / 2, 0.0]),
        ([0.35, -0.2, -0.28], [0.0, -np.pi / 2, 0.0]),
    ]
    for p, e in t_ik:
        cat.append(("r_arm", p, e))
    # README.md:73-74 and src/benchmark/ik_benchmarks.py:13-14
    cat.append(("r_arm", [0.55, -0.3, -0.15], [0, -np.pi / 2, 0]))
    cat.append(("r_arm", [0.3, -0.1, 0.1], [d2r(20), d2r(-50), d2r(20)]))
    # src/example/test_go_to.py:169-213 labelled poses (r list; l list is its mirror)
    go_to_r = [
        ([0.0001, -0.2, -0.6599], [0, 0, 0]),
        ([0.38, -0.2, -0.28], [0, -np.pi / 2, 0]),
        ([0.66, -0.2, -0.0], [0, -np.pi / 2, 0]),
        ([0.30, -0.2, -0.28], [0.0, 0.0, np.pi / 3]),
        ([0.0, -0.85, -0.0], [-np.pi / 2, 0, 0]),
        ([0.0, -0.58, -0.28], [-np.pi / 2, -np.pi / 2, 0]),
        ([0.15, 0.35, -0.10], [np.pi / 3, -np.pi / 2, 0]),
        ([0.10, 0.20, -0.22], [np.pi / 3, -np.pi / 2, 0]),
        ([0.0, -0.2, -0.66], [0.0, 0.0, -np.pi / 3]),
        ([0.001, -0.2, -0.68], [0.0, 0.0, -np.pi / 3]),
        ([0.001, -0.2, -0.659], [0.0, np.pi / 2, 0.0]),
        ([0.38, -0.2, -0.28], [0.0, np.pi / 2, 0.0]),
        ([0.1, -0.2, 0.0], [0.0, np.pi, 0.0]),
        ([0.38, -0.2, -0.28], [0.0, 0.0, 0.0]),
        ([0.1, 0.2, -0.1], [0.0, -np.pi / 2, np.pi / 2]),
        ([0.0, -0.2, -0.66], [0, 0, 0]),
    ]
    for p, e in go_to_r:
        cat.append(("r_arm", p, e))
    for p, e in go_to_r:
        pl, el = mirror_pose(np.array(p, dtype=float), np.array(e, dtype=float))
        cat.append(("l_arm", list(pl), list(el)))
    # mirrors of the test_ik / README / benchmark poses for the l arm
    for p, e in t_ik + [([0.55, -0.3, -0.15], [0, -np.pi / 2, 0]), ([0.3, -0.1, 0.1], [d2r(20), d2r(-50), d2r(20)])]:
        pl, el = mirror_pose(np.array(p, dtype=float), np.array(e, dtype=float))
        cat.append(("l_arm", list(pl), list(el)))
    # geometric edge cases: fully extended arm along +x, min-distance reduce, backward shift of wrist
    cat.append(("r_arm", [0.66, -0.2, 0.0], [0.0, -np.pi / 2, 0.0]))
    cat.append(("r_arm", [0.2, -0.2, -0.05], [0.0, -np.pi / 2, 0.0]))
    cat.append(("r_arm", [0.15, -0.25, -0.1], [0.0, -np.pi / 2, 0.3]))
    cat.append(("r_arm", [0.05, -0.2, -0.5], [0.0, 0.3, 0.0]))
    cat.append(("r_arm", [0.03, -0.3, -0.45], [0.2, 0.6, -0.1]))
    cat.append(("l_arm", [0.2, 0.2, -0.05], [0.0, -np.pi / 2, 0.0]))
    cat.append(("l_arm", [0.05, 0.2, -0.5], [0.0, 0.3, 0.0]))
    return cat


def control_call(ctrl, arm, M, nb, mode, current_joints=None, preferred_theta=None):
    ctrl.nb_search_points = nb
    kw = {}
    if current_joints is not None:
        kw["current_joints"] = list(current_joints)
    if preferred_theta is not None:
        kw["preferred_theta"] = preferred_theta
    j, ok, st = quiet(ctrl.symbolic_inverse_kinematics, arm, M, "discrete", constrained_mode=mode, **kw)
    assert not ctrl.emergency_stop
    return np.array(j, dtype=float), np.uint8(bool(ok)), np.uint8(STATE_CODES[st])


def gen_catalogue(out):
    cat = catalogue()
    solvers = {(arm, so): make_solver(arm, so) for arm in ARMS for so in (0.03, -1.01)}
    ctrl = quiet(ControlIK, urdf_path="../config_files/reachy2.urdf")
    rows03, rows101 = [], []
    arm_id, pos, eul, Ms = [], [], [], []
    cj = {k: [] for k in ("u20", "u64", "l20", "l64")}
    cf = {k: [] for k in cj}
    cs = {k: [] for k in cj}
    for arm, p, e in cat:
        arm_id.append(ARMS.index(arm))
        pos.append(np.array(p, dtype=float))
        eul.append(np.array(e, dtype=float))
        rows03.append(solve_symbolic(solvers[(arm, 0.03)], p, e))
        rows101.append(solve_symbolic(solvers[(arm, -1.01)], p, e))
        M = pose_to_matrix(p, e)
        Ms.append(M)
        for key, nb, mode in (("u20", 20, "unconstrained"), ("u64", 64, "unconstrained"),
                              ("l20", 20, "low_elbow"), ("l64", 64, "low_elbow")):
            j, ok, st = control_call(ctrl, arm, M, nb, mode)
            cj[key].append(j)
            cf[key].append(ok)
            cs[key].append(st)
    data = {"arm": np.array(arm_id, dtype=np.uint8), "pos": np.array(pos), "eul": np.array(eul), "M": np.array(Ms)}
    for k, v in stack(rows03).items():
        data["so003_" + k] = v
    for k, v in stack(rows101).items():
        data["so101_" + k] = v
    for k in cj:
        data[f"ctrl_{k}_joints"] = np.array(cj[k])
        data[f"ctrl_{k}_reachable"] = np.array(cf[k], dtype=np.uint8)
        data[f"ctrl_{k}_state"] = np.array(cs[k], dtype=np.uint8)
    np.savez_compressed(os.path.join(out, "g1_catalogue.npz"), **data)


# ----------------------------------------------------------------------------------------
# G2: random sweep, all outcomes
# ----------------------------------------------------------------------------------------
def gen_sweep(out, n=20000):
    rng = np.random.default_rng(0)
    data = {}
    for arm in ARMS:
        solver = make_solver(arm, 0.03)
        pos, eul = random_poses(rng, arm, n)
        rows = [solve_symbolic(solver, p, e) for p, e in zip(pos, eul)]
        data[f"{arm}_pos"] = pos
        data[f"{arm}_eul"] = eul
        for k, v in stack(rows).items():
            data[f"{arm}_{k}"] = v
    np.savez_compressed(os.path.join(out, "g2_sweep.npz"), **data)


# ----------------------------------------------------------------------------------------
# G3: reachable only, two theta policies, two singularity offsets, prev joints
# ----------------------------------------------------------------------------------------
def gen_reachable(out, n=4096):
    rng = np.random.default_rng(1)
    data = {}
    for arm in ARMS:
        base = make_solver(arm, 0.03)
        pos, eul = reachable_poses(rng, base, arm, n)
        tu = rng.uniform(0.0, 1.0, size=n)
        data[f"{arm}_pos"] = pos
        data[f"{arm}_eul"] = eul
        data[f"{arm}_theta_u"] = tu
        for tag, so in (("so003", 0.03), ("so101", -1.01)):
            solver = make_solver(arm, so)
            rows0 = [solve_symbolic(solver, p, e) for p, e in zip(pos, eul)]
            rows1 = [solve_symbolic(solver, p, e, theta_u=u) for p, e, u in zip(pos, eul, tu)]
            for k, v in stack(rows0).items():
                data[f"{arm}_{tag}_i0_{k}"] = v
            for k, v in stack(rows1).items():
                data[f"{arm}_{tag}_in_{k}"] = v
    np.savez_compressed(os.path.join(out, "g3_reachable.npz"), **data)


# ----------------------------------------------------------------------------------------
# G4: ControlIK discrete
# ----------------------------------------------------------------------------------------
def gen_control(out, n_uniform=1024, n_reach=1024, n_var=512):
    rng = np.random.default_rng(2)
    data = {}
    for dvt_tag, is_dvt in (("std", False), ("dvt", True)):
        ctrl = quiet(ControlIK, urdf_path="../config_files/reachy2.urdf", is_dvt=is_dvt)
        for arm in ARMS:
            solver = ctrl.symbolic_ik_solver[arm]
            pu, eu = random_poses(rng, arm, n_uniform)
            pr, er = reachable_poses(rng, solver, arm, n_reach)
            pos = np.concatenate([pu, pr])
            eul = np.concatenate([eu, er])
            Ms = np.array([pose_to_matrix(p, e) for p, e in zip(pos, eul)])
            pre = f"{dvt_tag}_{arm}_"
            data[pre + "M"] = Ms
            combos = [("u20", 20, "unconstrained"), ("u64", 64, "unconstrained"),
                      ("l20", 20, "low_elbow"), ("l64", 64, "low_elbow")]
            if is_dvt:
                combos = combos[:2]
            for key, nb, mode in combos:
                res = [control_call(ctrl, arm, M, nb, mode) for M in Ms]
                data[pre + key + "_joints"] = np.array([r[0] for r in res])
                data[pre + key + "_reachable"] = np.array([r[1] for r in res], dtype=np.uint8)
                data[pre + key + "_state"] = np.array([r[2] for r in res], dtype=np.uint8)
            if not is_dvt:
                # argument variations: explicit current_joints and preferred_theta (r-arm convention)
                idx = rng.choice(len(Ms), size=n_var, replace=False)
                curj = rng.uniform(-1.0, 1.0, size=(n_var, 7))
                pth = rng.uniform(-np.pi, np.pi, size=n_var)
                res = [control_call(ctrl, arm, Ms[i], 20, "unconstrained", current_joints=c, preferred_theta=t)
                       for i, c, t in zip(idx, curj, pth)]
                data[pre + "var_idx"] = idx.astype(np.int64)
                data[pre + "var_current_joints"] = curj
                data[pre + "var_preferred_theta"] = pth
                data[pre + "var_joints"] = np.array([r[0] for r in res])
                data[pre + "var_reachable"] = np.array([r[1] for r in res], dtype=np.uint8)
                data[pre + "var_state"] = np.array([r[2] for r in res], dtype=np.uint8)
    np.savez_compressed(os.path.join(out, "g4_control_discrete.npz"), **data)


# ----------------------------------------------------------------------------------------
# G5: elbow positions + is_reachable_no_limits (helpers used by theta search / continuous)
# ----------------------------------------------------------------------------------------
def gen_helpers(out, n=2048):
    rng = np.random.default_rng(3)
    data = {}
    for arm in ARMS:
        solver = make_solver(arm, 0.03)
        pr, er = reachable_poses(rng, solver, arm, n // 2)
        pu, eu = random_poses(rng, arm, n // 2)
        pos = np.concatenate([pr, pu])
        eul = np.concatenate([er, eu])
        thetas = rng.uniform(-np.pi, np.pi, size=(len(pos), 4))
        elb = np.full((len(pos), 4, 3), NAN)
        nl_joints = np.full((len(pos), 7), NAN)
        nl_elbow = np.full((len(pos), 3), NAN)
        nl_ok = np.zeros(len(pos), dtype=np.uint8)
        for i, (p, e) in enumerate(zip(pos, eul)):
            pose = np.array([p, e])
            ok, _, _, _ = solver.is_reachable(pose)
            if ok:
                for k in range(4):
                    elb[i, k] = solver.get_elbow_position(thetas[i, k])[:3]
            ok2, itv, fn = solver.is_reachable_no_limits(pose)
            nl_ok[i] = bool(ok2)
            if ok2:
                j, el = fn(thetas[i, 0])
                nl_joints[i] = j
                nl_elbow[i] = el[:3]
        data[f"{arm}_pos"] = pos
        data[f"{arm}_eul"] = eul
        data[f"{arm}_thetas"] = thetas
        data[f"{arm}_elbow_at_theta"] = elb
        data[f"{arm}_nolimits_ok"] = nl_ok
        data[f"{arm}_nolimits_joints"] = nl_joints
        data[f"{arm}_nolimits_elbow"] = nl_elbow
    np.savez_compressed(os.path.join(out, "g5_helpers.npz"), **data)


# ----------------------------------------------------------------------------------------
# G6: ControlIK continuous over pose trajectories, deterministic fake clock
# ----------------------------------------------------------------------------------------
class FakeClock:
    def __init__(self, t0=1000.0):
        self.t = t0

    def time(self):
        return self.t


def trajectory_pose(t, arm):
    """Task-space generator shaped like tests/test_sdk.py:38-63 (data, not code)."""
    x0, y0, z0 = 0.65, -0.2, 0.0
    r0, p0, w0 = 0.0, -np.pi / 2, 0.0
    amp = [0.35, 0.35, 0.35, np.pi / 6, np.pi / 6, np.pi / 6]
    freq = [0.6, 0.34, 0.78, 0.18, 0.31, 0.47]
    v = [c + a * np.sin(f * t) for c, a, f in zip((x0, y0, z0, r0, p0, w0), amp, freq)]
    pos, eul = np.array(v[:3]), np.array(v[3:])
    if arm == "l_arm":
        pos, eul = mirror_pose(pos, eul)
    return pos, eul


def gen_continuous(out, n_traj=6, n_steps=400):
    rng = np.random.default_rng(4)
    data = {}
    real_time = ref_control_mod.time
    for arm in ARMS:
        phases = rng.uniform(0.0, 40.0, size=n_traj)
        Ms = np.zeros((n_traj, n_steps, 4, 4))
        J = np.zeros((n_traj, n_steps, 7))
        F = np.zeros((n_traj, n_steps), dtype=np.uint8)
        S = np.zeros((n_traj, n_steps), dtype=np.uint8)
        TH = np.zeros((n_traj, n_steps))
        ES = np.zeros((n_traj, n_steps), dtype=np.uint8)
        for k in range(n_traj):
            clock = FakeClock()
            ref_control_mod.time = clock
            try:
                ctrl = quiet(ControlIK, urdf_path="../config_files/reachy2.urdf")
                for i in range(n_steps):
                    t = i / 120.0 + 11.0 + phases[k]
                    pos, eul = trajectory_pose(t, arm)
                    M = pose_to_matrix(pos, eul)
                    Ms[k, i] = M
                    clock.t += 1.0 / 120.0
                    j, ok, st = quiet(ctrl.symbolic_inverse_kinematics, arm, M, "continuous", d_theta_max=0.01)
                    J[k, i] = np.array(j, dtype=float)
                    F[k, i] = bool(ok)
                    S[k, i] = STATE_CODES.get(st, 255)
                    TH[k, i] = ctrl.previous_theta[arm]
                    ES[k, i] = bool(ctrl.emergency_stop)
            finally:
                ref_control_mod.time = real_time
        data[f"{arm}_phase"] = phases
        data[f"{arm}_M"] = Ms
        data[f"{arm}_joints"] = J
        data[f"{arm}_reachable"] = F
        data[f"{arm}_state"] = S
        data[f"{arm}_previous_theta"] = TH
        data[f"{arm}_emergency_stop"] = ES
    np.savez_compressed(os.path.join(out, "g6_control_continuous.npz"), **data)


def gen_continuous_start(out, n_traj=8, n_steps=150):
    """G7: like G6 but every trajectory starts from an explicit, generic (current_joints, current_pose) pair, so the
    start-up ternary search (utils.py:267-331) is not sitting on the exact tie it hits for the constructor's default
    arms-along-the-body configuration."""
    rng = np.random.default_rng(6)
    data = {}
    real_time = ref_control_mod.time
    for arm in ARMS:
        phases = rng.uniform(0.0, 40.0, size=n_traj)
        cur_j = rng.uniform(-0.6, 0.6, size=(n_traj, 7))
        Ms = np.zeros((n_traj, n_steps, 4, 4))
        P0 = np.zeros((n_traj, 4, 4))
        J = np.zeros((n_traj, n_steps, 7))
        F = np.zeros((n_traj, n_steps), dtype=np.uint8)
        S = np.zeros((n_traj, n_steps), dtype=np.uint8)
        TH = np.zeros((n_traj, n_steps))
        for k in range(n_traj):
            clock = FakeClock()
            ref_control_mod.time = clock
            try:
                ctrl = quiet(ControlIK, urdf_path="../config_files/reachy2.urdf", is_dvt=(k % 2 == 1))
                p0, e0 = trajectory_pose(11.0 + phases[k] - 0.3, arm)
                P0[k] = pose_to_matrix(p0 * np.array([0.8, 1.0, 1.0]) + rng.uniform(-0.02, 0.02, 3), e0 + rng.uniform(-0.1, 0.1, 3))
                for i in range(n_steps):
                    t = i / 120.0 + 11.0 + phases[k]
                    pos, eul = trajectory_pose(t, arm)
                    M = pose_to_matrix(pos, eul)
                    Ms[k, i] = M
                    clock.t += 1.0 / 120.0
                    kw = {}
                    if i == 0:
                        kw = dict(current_joints=list(cur_j[k]), current_pose=P0[k])
                    j, ok, st = quiet(ctrl.symbolic_inverse_kinematics, arm, M, "continuous", d_theta_max=0.01, **kw)
                    J[k, i] = np.array(j, dtype=float)
                    F[k, i] = bool(ok)
                    S[k, i] = STATE_CODES.get(st, 8)
                    TH[k, i] = ctrl.previous_theta[arm]
            finally:
                ref_control_mod.time = real_time
        data[f"{arm}_M"] = Ms
        data[f"{arm}_start_pose"] = P0
        data[f"{arm}_start_joints"] = cur_j
        data[f"{arm}_is_dvt"] = (np.arange(n_traj) % 2 == 1).astype(np.uint8)
        data[f"{arm}_joints"] = J
        data[f"{arm}_reachable"] = F
        data[f"{arm}_state"] = S
        data[f"{arm}_previous_theta"] = TH
    np.savez_compressed(os.path.join(out, "g7_control_continuous_start.npz"), **data)


# ----------------------------------------------------------------------------------------
# G8: goal matrices at the edges of the matrix -> Euler conversion: proper rotations, gimbal lock
# (pitch at and around +-pi/2, where as_euler sets the third angle to 0), matrices that are not quite orthonormal
# (from_matrix's quaternion normalisation decides), near-identity matrices (np.allclose shortcut, control_ik.py:212).
# Records utils.get_euler_from_homogeneous_matrix and the ControlIK discrete result for each.
# ----------------------------------------------------------------------------------------
def gen_matrix_edges(out, n=384):
    import warnings

    from reachy2_symbolic_ik.utils import get_euler_from_homogeneous_matrix

    rng = np.random.default_rng(8)
    data = {}
    ctrl = quiet(ControlIK, urdf_path="../config_files/reachy2.urdf")
    for arm in ARMS:
        solver = ctrl.symbolic_ik_solver[arm]
        pos, eul = reachable_poses(rng, solver, arm, 4 * n)
        kinds = {}
        kinds["proper"] = np.array([pose_to_matrix(p, e) for p, e in zip(pos[:n], eul[:n])])
        # gimbal lock: pitch = +-(pi/2 - delta), delta from 0 to 3e-7 (the convention switches at 1e-7 on the
        # quaternion-derived middle angle)
        deltas = np.array([0.0, 1e-12, 1e-9, 2e-8, 6e-8, 9e-8, 1.1e-7, 1.5e-7, 3e-7])
        eg = eul[n:2 * n].copy()
        sign = np.where(rng.uniform(size=n) < 0.5, -1.0, 1.0)
        eg[:, 1] = sign * (np.pi / 2 - deltas[np.arange(n) % len(deltas)])
        kinds["gimbal"] = np.array([pose_to_matrix(p, e) for p, e in zip(pos[n:2 * n], eg)])
        # not orthonormal: every rotation entry scaled by 1 + U(-1e-3, 1e-3)
        Mn = np.array([pose_to_matrix(p, e) for p, e in zip(pos[2 * n:3 * n], eul[2 * n:3 * n])])
        Mn[:, :3, :3] *= 1.0 + rng.uniform(-1e-3, 1e-3, size=(n, 3, 3))
        kinds["skewed"] = Mn
        # near the identity: rotations of 1e-9 .. 1e-4 rad around random axes (the allclose shortcut ends near 1e-5)
        ang = 10.0 ** rng.uniform(-9, -4, size=n)
        axis = rng.normal(size=(n, 3))
        axis /= np.linalg.norm(axis, axis=1, keepdims=True)
        Mi = np.tile(np.eye(4), (n, 1, 1))
        Mi[:, :3, :3] = R.from_rotvec(axis * ang[:, None]).as_matrix()
        Mi[:, :3, 3] = pos[3 * n:4 * n]
        kinds["near_identity"] = Mi
        for kind, Ms in kinds.items():
            pre = f"{arm}_{kind}_"
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")  # scipy's gimbal-lock warning
                eu = np.array([get_euler_from_homogeneous_matrix(M)[1] for M in Ms])
                res = [control_call(ctrl, arm, M, 20, "unconstrained") for M in Ms]
            data[pre + "M"] = Ms
            data[pre + "euler"] = eu
            data[pre + "joints"] = np.array([r[0] for r in res])
            data[pre + "reachable"] = np.array([r[1] for r in res], dtype=np.uint8)
            data[pre + "state"] = np.array([r[2] for r in res], dtype=np.uint8)
    np.savez_compressed(os.path.join(out, "g8_matrix_edges.npz"), **data)


# ----------------------------------------------------------------------------------------
# G9: non-default geometry.  Every other fixture uses the reference's default arm (tip = [0, 0, 0.10], u = f = 0.28,
# default limits), so the general formulas (tip with x / y components, unequal segment lengths, other shoulder
# offsets, other elbow / wrist / backward limits and singularity plane) are pinned here.
# ----------------------------------------------------------------------------------------
CUSTOM_GEOMETRY = {
    "ik_parameters": {
        "r_shoulder_position": np.array([0.012, -0.185, 0.021]),
        "r_shoulder_orientation": [-11.0, 3.5, 7.0],
        "r_upper_arm_size": np.float64(0.305),
        "r_forearm_size": np.float64(0.262),
        "r_tip_position": np.array([0.013, -0.008, 0.094]),
        "l_shoulder_position": np.array([0.012, 0.185, 0.021]),
        "l_shoulder_orientation": [11.0, 3.5, -7.0],
        "l_upper_arm_size": np.float64(0.305),
        "l_forearm_size": np.float64(0.262),
        "l_tip_position": np.array([0.013, 0.008, 0.094]),
    },
    "elbow_limit": 115,
    "wrist_limit": np.float64(38.0),
    "backward_limit": 0.035,
    "singularity_offset": 0.05,
    "singularity_limit_coeff": 0.8,
}


def gen_custom_geometry(out, n_sweep=6000, n_reach=1536):
    rng = np.random.default_rng(9)
    data = {}
    for arm in ARMS:
        solver = quiet(SymbolicIK, arm=arm, **CUSTOM_GEOMETRY)
        for f in ("gripper_size", "max_arm_length", "shoulder_wrist_min_distance", "elbow_singularity_position",
                  "wrist_singularity_position"):
            data[f"{arm}_const_{f}"] = np.asarray(getattr(solver, f), dtype=float)
        sh = CUSTOM_GEOMETRY["ik_parameters"][f"{arm[0]}_shoulder_position"]
        pos = sh + rng.uniform(-0.7, 0.7, size=(n_sweep, 3))
        eul = rng.uniform(-np.pi, np.pi, size=(n_sweep, 3))
        rows = [solve_symbolic(solver, p, e) for p, e in zip(pos, eul)]
        data[f"{arm}_sweep_pos"] = pos
        data[f"{arm}_sweep_eul"] = eul
        for k, v in stack(rows).items():
            data[f"{arm}_sweep_{k}"] = v
        P, E = [], []
        while len(P) < n_reach:
            pp = sh + rng.uniform(-0.7, 0.7, size=(4096, 3))
            ee = rng.uniform(-np.pi, np.pi, size=(4096, 3))
            for p, e in zip(pp, ee):
                if solver.is_reachable(np.array([p, e]))[0]:
                    P.append(p)
                    E.append(e)
                    if len(P) == n_reach:
                        break
        P, E = np.array(P), np.array(E)
        tu = rng.uniform(0.0, 1.0, size=n_reach)
        data[f"{arm}_reach_pos"] = P
        data[f"{arm}_reach_eul"] = E
        data[f"{arm}_reach_theta_u"] = tu
        for k, v in stack([solve_symbolic(solver, p, e) for p, e in zip(P, E)]).items():
            data[f"{arm}_reach_i0_{k}"] = v
        for k, v in stack([solve_symbolic(solver, p, e, theta_u=u) for p, e, u in zip(P, E, tu)]).items():
            data[f"{arm}_reach_in_{k}"] = v
    np.savez_compressed(os.path.join(out, "g9_custom_geometry.npz"), **data)


# ----------------------------------------------------------------------------------------
# G10: ControlIK built from a URDF that is NOT the Reachy 2 one (tests/golden/custom_arm.urdf, test data written for
# this repo): pins URDF -> parameters -> constants -> discrete / continuous control for a non-default geometry.
# ----------------------------------------------------------------------------------------
def gen_custom_urdf_control(out, n=512, n_traj=4, n_steps=200):
    urdf = open(os.path.join(out, "custom_arm.urdf")).read()
    rng = np.random.default_rng(10)
    data = {}
    real_time = ref_control_mod.time
    for arm in ARMS:
        ctrl = quiet(ControlIK, urdf=urdf)
        solver = ctrl.symbolic_ik_solver[arm]
        for f in ("shoulder_position", "shoulder_orientation_offset", "upper_arm_size", "forearm_size", "tip_position"):
            data[f"{arm}_param_{f}"] = np.asarray(getattr(solver, f), dtype=float)
        sh = np.asarray(solver.shoulder_position, dtype=float)
        P, E = [], []
        while len(P) < n // 2:
            pp = sh + rng.uniform(-0.7, 0.7, size=(2048, 3))
            ee = rng.uniform(-np.pi, np.pi, size=(2048, 3))
            for p, e in zip(pp, ee):
                if solver.is_reachable(np.array([p, e]))[0]:
                    P.append(p)
                    E.append(e)
                    if len(P) == n // 2:
                        break
        pos = np.concatenate([np.array(P), sh + rng.uniform(-0.7, 0.7, size=(n - n // 2, 3))])
        eul = np.concatenate([np.array(E), rng.uniform(-np.pi, np.pi, size=(n - n // 2, 3))])
        Ms = np.array([pose_to_matrix(p, e) for p, e in zip(pos, eul)])
        data[f"{arm}_M"] = Ms
        for key, nb, mode in (("u20", 20, "unconstrained"), ("l64", 64, "low_elbow")):
            res = [control_call(ctrl, arm, M, nb, mode) for M in Ms]
            data[f"{arm}_{key}_joints"] = np.array([r[0] for r in res])
            data[f"{arm}_{key}_reachable"] = np.array([r[1] for r in res], dtype=np.uint8)
            data[f"{arm}_{key}_state"] = np.array([r[2] for r in res], dtype=np.uint8)
        # continuous mode from explicit generic starts (see G7)
        phases = rng.uniform(0.0, 40.0, size=n_traj)
        cur_j = rng.uniform(-0.6, 0.6, size=(n_traj, 7))
        TM = np.zeros((n_traj, n_steps, 4, 4))
        P0 = np.zeros((n_traj, 4, 4))
        J = np.zeros((n_traj, n_steps, 7))
        F = np.zeros((n_traj, n_steps), dtype=np.uint8)
        S = np.zeros((n_traj, n_steps), dtype=np.uint8)
        for k in range(n_traj):
            clock = FakeClock()
            ref_control_mod.time = clock
            try:
                c2 = quiet(ControlIK, urdf=urdf)
                p0, e0 = trajectory_pose(11.0 + phases[k] - 0.3, arm)
                P0[k] = pose_to_matrix(p0 * np.array([0.8, 1.0, 1.0]) + rng.uniform(-0.02, 0.02, 3), e0 + rng.uniform(-0.1, 0.1, 3))
                for i in range(n_steps):
                    pos_i, eul_i = trajectory_pose(i / 120.0 + 11.0 + phases[k], arm)
                    M = pose_to_matrix(pos_i, eul_i)
                    TM[k, i] = M
                    clock.t += 1.0 / 120.0
                    kw = dict(current_joints=list(cur_j[k]), current_pose=P0[k]) if i == 0 else {}
                    j, ok, st = quiet(c2.symbolic_inverse_kinematics, arm, M, "continuous", d_theta_max=0.01, **kw)
                    J[k, i] = np.array(j, dtype=float)
                    F[k, i] = bool(ok)
                    S[k, i] = STATE_CODES.get(st, 8)
            finally:
                ref_control_mod.time = real_time
        data[f"{arm}_traj_M"] = TM
        data[f"{arm}_traj_start_pose"] = P0
        data[f"{arm}_traj_start_joints"] = cur_j
        data[f"{arm}_traj_joints"] = J
        data[f"{arm}_traj_reachable"] = F
        data[f"{arm}_traj_state"] = S
    np.savez_compressed(os.path.join(out, "g10_custom_urdf_control.npz"), **data)


# ----------------------------------------------------------------------------------------
# G11: emergency stops and the diagnostics the reference reports for them
# (utils.multiturn_safety_check utils.py:535-568, utils.continuity_check utils.py:571-589,
#  ControlIK.symbolic_inverse_kinematics control_ik.py:196-210: the latched state returns emergency_state as `state`)
# ----------------------------------------------------------------------------------------
def _near_limit(jstar, sign, delta=0.1):
    """previous value of a joint such that allow_multiturn(jstar, prev) = prev + sign * delta lands beyond +-6 pi."""
    lim = 6 * np.pi
    if sign > 0:
        m = np.ceil((lim - jstar) / (2 * np.pi))
        return jstar + 2 * np.pi * m - delta
    m = np.ceil((lim + jstar) / (2 * np.pi))
    return jstar - 2 * np.pi * m + delta


def gen_emergency(out):
    rng = np.random.default_rng(11)
    data = {}
    real_time = ref_control_mod.time
    cases = [((0, +1),), ((0, -1),), ((2, +1),), ((2, -1),), ((6, +1),), ((6, -1),), ((0, -1), (6, +1)), ((0, +1), (2, -1), (6, -1))]
    for arm in ARMS:
        ai = ARMS.index(arm)
        # ---- discrete mode: previous_sol next to the +-6 pi multiturn limit
        D = dict(M=[], current_joints=[], joints1=[], ok1=[], state1=[], emergency_state=[], joints2=[], ok2=[], state2=[], cause=[])
        tries = 0
        for case in cases * 2:
            while True:
                tries += 1
                pos, eul = trajectory_pose(11.0 + rng.uniform(0, 40.0), arm)
                M = pose_to_matrix(pos, eul)
                c0 = quiet(ControlIK, urdf_path="../config_files/reachy2.urdf")
                jstar, ok, _ = quiet(c0.symbolic_inverse_kinematics, arm, M, "discrete")
                if ok:
                    break
            jstar = np.array(jstar, dtype=float)
            cur = [list(ref_default_joints(0)), list(ref_default_joints(1))]
            prev = jstar + rng.uniform(-0.05, 0.05, 7)
            cause = 0
            for (k, sign) in case:
                prev[k] = _near_limit(jstar[k], sign)
                cause |= {0: 1, 2: 2, 6: 4}[k]
            cur[ai] = list(prev)
            c = quiet(ControlIK, current_joints=cur, urdf_path="../config_files/reachy2.urdf")
            j1, ok1, st1 = quiet(c.symbolic_inverse_kinematics, arm, M, "discrete")
            assert c.emergency_stop, (arm, case)
            es = c.emergency_state
            j2, ok2, st2 = quiet(c.symbolic_inverse_kinematics, arm, M, "discrete")
            assert st2 == es and not ok2
            for key, val in (("M", M), ("current_joints", np.array(cur)), ("joints1", np.array(j1, dtype=float)), ("ok1", bool(ok1)),
                             ("state1", st1), ("emergency_state", es), ("joints2", np.array(j2, dtype=float)), ("ok2", bool(ok2)),
                             ("state2", st2), ("cause", cause)):
                D[key].append(val)
        for key, val in D.items():
            data[f"{arm}_discrete_{key}"] = np.array(val)
        # ---- continuous mode: a jump of the goal (continuity_check) and a start next to the multiturn limit
        n_traj, n_steps, jump_at = 6, 12, 6
        C = dict(M=np.zeros((n_traj, n_steps, 4, 4)), joints=np.zeros((n_traj, n_steps, 7)), ok=np.zeros((n_traj, n_steps), dtype=np.uint8),
                 state=[], emergency_state=[], start_joints=np.full((n_traj, 7), np.nan), control_type=[], estop=np.zeros((n_traj, n_steps), dtype=np.uint8),
                 previous_sol=np.zeros((n_traj, n_steps, 7)))
        for k in range(n_traj):
            for attempt in range(200):  # draw phases until the scenario really trips the emergency stop
                clock = FakeClock()
                ref_control_mod.time = clock
                try:
                    c = quiet(ControlIK, urdf_path="../config_files/reachy2.urdf")
                    phase = rng.uniform(0.0, 40.0)
                    states, ctypes, es_list = [], [], []
                    multiturn_start = k >= 4
                    if multiturn_start:
                        pos, eul = trajectory_pose(11.0 + phase, arm)
                        cprobe = quiet(ControlIK, urdf_path="../config_files/reachy2.urdf")
                        jstar, _, _ = quiet(cprobe.symbolic_inverse_kinematics, arm, pose_to_matrix(pos, eul), "continuous", d_theta_max=0.01)
                        start = np.array(jstar, dtype=float) + rng.uniform(-0.02, 0.02, 7)
                        which = 0 if k == 4 else 6
                        start[which] = _near_limit(float(jstar[which]), -1 if k == 4 else +1)
                        C["start_joints"][k] = start
                    for i in range(n_steps):
                        t = i / 120.0 + 11.0 + phase
                        if not multiturn_start and i >= jump_at:
                            t += 2.5 + k  # the goal jumps: some joint moves by more than its continuity threshold
                        pos, eul = trajectory_pose(t, arm)
                        M = pose_to_matrix(pos, eul)
                        clock.t += 1.0 / 120.0
                        ctype = "unfreeze" if i == n_steps - 3 else "continuous"
                        kw = dict(current_joints=list(C["start_joints"][k])) if (multiturn_start and i == 0) else {}
                        j, ok, st = quiet(c.symbolic_inverse_kinematics, arm, M, ctype, d_theta_max=0.01, **kw)
                        C["M"][k, i] = M
                        C["joints"][k, i] = np.array(j, dtype=float)
                        C["ok"][k, i] = bool(ok)
                        C["estop"][k, i] = bool(c.emergency_stop)
                        C["previous_sol"][k, i] = np.array(c.previous_sol[arm], dtype=float)
                        states.append(st)
                        ctypes.append(ctype)
                        es_list.append(c.emergency_state)
                finally:
                    ref_control_mod.time = real_time
                if C["estop"][k, : n_steps - 3].any():
                    break
            else:
                raise AssertionError((arm, k, "no emergency stop in 200 draws"))
            C["state"].append(states)
            C["control_type"].append(ctypes)
            C["emergency_state"].append(es_list)
        for key, val in C.items():
            data[f"{arm}_continuous_{key}"] = np.array(val)
    np.savez_compressed(os.path.join(out, "g11_emergency.npz"), **data)


# ----------------------------------------------------------------------------------------
# G12: ControlIK continuous in EVERY mode: constrained_mode x d_theta_max x preferred_theta argument x DVT, both arms
# (control_ik.py:225-252 interval_limit / l-arm mirror, :350-384 both branches, utils.py:93-112 limit_theta_to_interval,
#  :115-127 tend_to_preferred_theta with the ARGUMENT as goal, :220-264 get_best_continuous_theta2 with self.preferred_theta).
# The preferred_theta arguments 0.5 and 2.0 (r-arm convention; the reference mirrors them for the left arm) lie OUTSIDE
# both control intervals, next to one end each, so the unreachable stretches of the trajectory (66 % of the steps) pull
# theta out of the interval — limit_theta_to_interval snaps it to either end — and the reachable stretches pull it
# back in towards self.preferred_theta.  Two trajectories per combination: default start and explicit generic start.
# Also recorded: the RuntimeError of control_ik.py:385-387 (is_reachable_no_limits false), which needs a solver whose
# projection_margin is negative (a public attribute swap: ControlIK itself always builds solvers with 1e-8).
# ----------------------------------------------------------------------------------------
G12_MODES = ("unconstrained", "low_elbow")
G12_DTHETA = (0.01, 0.05, 0.4)
G12_PREFERRED = (-4 * np.pi / 6, 0.5, 2.0)


def gen_continuous_modes(out, n_steps=160):
    rng = np.random.default_rng(12)
    data = {}
    real_time = ref_control_mod.time
    for arm in ARMS:
        rows = []
        for mode_i, mode in enumerate(G12_MODES):
            for dth in G12_DTHETA:
                for pref in G12_PREFERRED:
                    for is_dvt in (0, 1):
                        for explicit in (0, 1):
                            rows.append((mode_i, dth, pref, is_dvt, explicit))
        n = len(rows)
        M12 = np.zeros((n, n_steps, 12))
        P0 = np.zeros((n, 4, 4))
        CJ = np.zeros((n, 7))
        J = np.zeros((n, n_steps, 7))
        F = np.zeros((n, n_steps), dtype=np.uint8)
        S = np.zeros((n, n_steps), dtype=np.uint8)
        TH = np.zeros((n, n_steps))
        ES = np.zeros((n, n_steps), dtype=np.uint8)
        for k, (mode_i, dth, pref, is_dvt, explicit) in enumerate(rows):
            clock = FakeClock()
            ref_control_mod.time = clock
            phase = rng.uniform(0.0, 40.0)
            try:
                ctrl = quiet(ControlIK, urdf_path="../config_files/reachy2.urdf", is_dvt=bool(is_dvt))
                ai = ARMS.index(arm)
                p0, e0 = trajectory_pose(11.0 + phase - 0.3, arm)
                P0[k] = pose_to_matrix(p0 * np.array([0.8, 1.0, 1.0]) + rng.uniform(-0.02, 0.02, 3), e0 + rng.uniform(-0.1, 0.1, 3))
                CJ[k] = rng.uniform(-0.6, 0.6, size=7)
                if not explicit:
                    P0[k] = np.asarray(ctrl.previous_pose[arm], dtype=float)
                    CJ[k] = ref_default_joints(ai)
                for i in range(n_steps):
                    pos, eul = trajectory_pose(i / 120.0 + 11.0 + phase, arm)
                    M = pose_to_matrix(pos, eul)
                    M12[k, i, :9] = M[:3, :3].reshape(9)
                    M12[k, i, 9:] = M[:3, 3]
                    clock.t += 1.0 / 120.0
                    kw = dict(current_joints=list(CJ[k]), current_pose=P0[k]) if (i == 0 and explicit) else {}
                    j, ok, st = quiet(ctrl.symbolic_inverse_kinematics, arm, M, "continuous", constrained_mode=G12_MODES[mode_i],
                                      d_theta_max=dth, preferred_theta=pref, **kw)
                    J[k, i] = np.array(j, dtype=float)
                    F[k, i] = bool(ok)
                    S[k, i] = STATE_CODES.get(st, 255)
                    TH[k, i] = ctrl.previous_theta[arm]
                    ES[k, i] = bool(ctrl.emergency_stop)
            finally:
                ref_control_mod.time = real_time
        rows = np.array(rows, dtype=float)
        data[f"{arm}_mode"] = rows[:, 0].astype(np.uint8)
        data[f"{arm}_d_theta_max"] = rows[:, 1]
        data[f"{arm}_preferred_theta"] = rows[:, 2]
        data[f"{arm}_is_dvt"] = rows[:, 3].astype(np.uint8)
        data[f"{arm}_explicit_start"] = rows[:, 4].astype(np.uint8)
        data[f"{arm}_M12"] = M12
        data[f"{arm}_start_pose"] = P0
        data[f"{arm}_start_joints"] = CJ
        data[f"{arm}_joints"] = J
        data[f"{arm}_reachable"] = F
        data[f"{arm}_state"] = S
        data[f"{arm}_previous_theta"] = TH
        data[f"{arm}_emergency_stop"] = ES
        # the deliberate crash of control_ik.py:385-387
        ref_control_mod.time = FakeClock()
        try:
            ctrl = quiet(ControlIK, urdf_path="../config_files/reachy2.urdf")
            ctrl.symbolic_ik_solver[arm] = quiet(SymbolicIK, arm=arm, projection_margin=-1e-3, singularity_offset=-1.01,
                                                 wrist_limit=np.rad2deg(ctrl.orbita3D_max_angle))
            far = np.array([0.9, SHOULDER[arm][1], 0.1])       # out of reach: pulled back with the negative margin
            near = np.array([0.45, SHOULDER[arm][1], -0.1])    # reachable: no crash
            raised, joints = [], []
            start = pose_to_matrix(near, np.array([0.0, -np.pi / 2, 0.0]))  # (the default start pose is itself on the rim)
            for p in (near, far, near):
                try:
                    j, ok, st = quiet(ctrl.symbolic_inverse_kinematics, arm, pose_to_matrix(p, np.array([0.0, -np.pi / 2, 0.0])),
                                      "continuous", current_joints=list(ref_default_joints(ARMS.index(arm))), current_pose=start)
                    raised.append(("", ""))
                    joints.append(np.array(j, dtype=float))
                except Exception as e:  # noqa: BLE001 - the type and text are the data
                    raised.append((type(e).__name__, str(e)))
                    joints.append(np.full(7, NAN))
            data[f"{arm}_crash_positions"] = np.array([near, far, near])
            data[f"{arm}_crash_raised"] = np.array(raised)
            data[f"{arm}_crash_joints"] = np.array(joints)
        finally:
            ref_control_mod.time = real_time
    np.savez_compressed(os.path.join(out, "g12_control_continuous_modes.npz"), **data)


# ----------------------------------------------------------------------------------------
# G13: goals that are not numbers — what the reference does with a NaN / an infinity in the pose or the matrix
# (include/rsik.h "Rows that are not numbers": where it raises, the build reports RSIK_STATE_INVALID_INPUT)
# ----------------------------------------------------------------------------------------
OUTCOME_RETURNED, OUTCOME_LINALG, OUTCOME_VALUE, OUTCOME_OTHER, OUTCOME_HANGS = 0, 1, 2, 3, 4


def _outcome(fn, forked=False, limit=8.0):
    """(outcome code, whatever fn returned or None).  LAPACK prints its own complaints about NaN arguments to the C stderr: not
    silenced, harmless.  forked: the call is made in a forked child with a time limit — for an infinite entry in the rotation
    scipy's Rotation.from_matrix may never come back from LAPACK (observed: R[0,0] = +inf), which no Python signal interrupts."""
    import warnings

    def run():
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                return OUTCOME_RETURNED, quiet(fn)
        except np.linalg.LinAlgError:
            return OUTCOME_LINALG, None
        except ValueError:
            return OUTCOME_VALUE, None
        except Exception:  # noqa: BLE001
            return OUTCOME_OTHER, None

    if not forked:
        return run()
    import multiprocessing as mp

    ctx = mp.get_context("fork")
    rx, tx = ctx.Pipe(duplex=False)

    def child():
        tx.send(run())
        tx.close()

    proc = ctx.Process(target=child)
    proc.start()
    got = rx.recv() if rx.poll(limit) else (OUTCOME_HANGS, None)
    if proc.is_alive() and got[0] == OUTCOME_HANGS:
        proc.kill()
    proc.join()
    return got


def gen_hostile(out, n_steps=48):
    poisons = [NAN, float("inf"), float("-inf")]
    bases = [([0.55, -0.3, -0.15], [0.0, -np.pi / 2, 0.0]), ([0.38, -0.2, -0.28], [0.3, -1.2, 0.2]),
             ([0.1, -0.2, 0.1], [0.0, -np.pi / 2, 0.0]), ([0.9, -0.2, 0.0], [0.0, -np.pi / 2, 0.0])]
    data = {}
    # (a) SymbolicIK.is_reachable (+ get_joints(interval[0]) when it says reachable)
    rows = {k: [] for k in ("arm", "pos", "eul", "outcome", "state", "reachable")}
    for ai, arm in enumerate(ARMS):
        solver = make_solver(arm)
        for bp, be in bases:
            if arm == "l_arm":
                bp, be = mirror_pose(np.array(bp), np.array(be))
            for comp in range(6):
                for v in poisons:
                    pos, eul = np.array(bp, dtype=float), np.array(be, dtype=float)
                    (pos if comp < 3 else eul)[comp % 3] = v

                    def call(pos=pos, eul=eul):
                        ok, interval, fn, state = solver.is_reachable(np.array([pos, eul]))
                        if ok:
                            fn(interval[0])
                        return bool(ok), state

                    oc, ret = _outcome(call, forked=True)
                    rows["arm"].append(ai)
                    rows["pos"].append(pos)
                    rows["eul"].append(eul)
                    rows["outcome"].append(oc)
                    rows["state"].append(STATE_CODES[ret[1]] if ret else 255)
                    rows["reachable"].append(int(ret[0]) if ret else 0)
    data["sym_arm"] = np.array(rows["arm"], dtype=np.uint8)
    data["sym_pos"], data["sym_eul"] = np.array(rows["pos"]), np.array(rows["eul"])
    data["sym_outcome"] = np.array(rows["outcome"], dtype=np.uint8)
    data["sym_state"] = np.array(rows["state"], dtype=np.uint8)
    data["sym_reachable"] = np.array(rows["reachable"], dtype=np.uint8)
    # (b) ControlIK discrete: every one of the twelve entries of the goal matrix, NaN / +inf / -inf
    Ms, ocs, sts, arms = [], [], [], []
    for ai, arm in enumerate(ARMS):
        for bp, be in bases[:2]:
            if arm == "l_arm":
                bp, be = mirror_pose(np.array(bp), np.array(be))
            for r_ in range(3):
                for c_ in range(4):
                    for v in poisons:
                        M = pose_to_matrix(np.array(bp, dtype=float), np.array(be, dtype=float))
                        M[r_, c_] = v
                        ctrl = quiet(ControlIK, urdf_path="../config_files/reachy2.urdf")
                        oc, ret = _outcome(lambda M=M, ctrl=ctrl: ctrl.symbolic_inverse_kinematics(arm, M, "discrete"), forked=True)
                        Ms.append(M)
                        ocs.append(oc)
                        sts.append(STATE_CODES.get(ret[2], 255) if ret else 255)
                        # (an infinite translation: "Pose out of reach" / "Backward pose" and current_joints, no exception)
                        arms.append(ai)
    data["disc_arm"] = np.array(arms, dtype=np.uint8)
    data["disc_M"] = np.array(Ms)
    data["disc_outcome"] = np.array(ocs, dtype=np.uint8)
    data["disc_state"] = np.array(sts, dtype=np.uint8)
    # (c) ControlIK continuous: a trajectory with some goals that are not numbers; the caller catches the exception and goes on
    # with the next goal, as a control loop would.  Recorded: what every call did, and previous_theta / previous_sol after it.
    real_time = ref_control_mod.time
    # (only goals on which the reference raises whatever the arm: a NaN anywhere, -inf on the rotation's diagonal (ValueError).
    # Other infinities it may answer — an infinite translation with "Pose out of reach" and NaN joints that then ARE its
    # previous_sol, an infinite rotation entry with the solution for whatever rotation scipy's nearest-rotation step makes of it,
    # or not at all — section (b).)
    bad_steps = {7: (0, 3, NAN), 19: (1, 1, float("-inf")), 20: (2, 2, NAN), 21: (1, 2, NAN), 40: (0, 0, NAN)}
    for arm in ARMS:
        Mt = np.zeros((n_steps, 4, 4))
        J = np.full((n_steps, 7), NAN)
        OC = np.zeros(n_steps, dtype=np.uint8)
        F = np.zeros(n_steps, dtype=np.uint8)
        S = np.full(n_steps, 255, dtype=np.uint8)
        TH = np.zeros(n_steps)
        PS = np.zeros((n_steps, 7))
        clock = FakeClock()
        ref_control_mod.time = clock
        try:
            ctrl = quiet(ControlIK, urdf_path="../config_files/reachy2.urdf")
            for i in range(n_steps):
                pos, eul = trajectory_pose(i / 120.0 + 13.0, arm)
                M = pose_to_matrix(pos, eul)
                if i in bad_steps:
                    r_, c_, v = bad_steps[i]
                    M[r_, c_] = v
                Mt[i] = M
                clock.t += 1.0 / 120.0
                oc, ret = _outcome(lambda M=M: ctrl.symbolic_inverse_kinematics(arm, M, "continuous", d_theta_max=0.01))
                OC[i] = oc
                if ret:
                    J[i], F[i], S[i] = np.array(ret[0], dtype=float), bool(ret[1]), STATE_CODES.get(ret[2], 255)
                TH[i] = ctrl.previous_theta[arm]
                PS[i] = np.array(ctrl.previous_sol[arm], dtype=float) if len(ctrl.previous_sol[arm]) == 7 else NAN
                assert not ctrl.emergency_stop and (oc != OUTCOME_RETURNED) == (i in bad_steps)
        finally:
            ref_control_mod.time = real_time
        data[f"cont_{arm}_M"], data[f"cont_{arm}_outcome"] = Mt, OC
        data[f"cont_{arm}_joints"], data[f"cont_{arm}_reachable"], data[f"cont_{arm}_state"] = J, F, S
        data[f"cont_{arm}_previous_theta"], data[f"cont_{arm}_previous_sol"] = TH, PS
    np.savez_compressed(os.path.join(out, "g13_hostile.npz"), **data)


# ----------------------------------------------------------------------------------------
# G14: BASELINE-scale digests.  The reference itself over the seeded generators of configs 2 and 3 at their full sizes
# (1 Mi poses per arm with every outcome; 256 Ki goal matrices through ControlIK discrete, 64-point grid): only digests,
# per-state counts and every 64th row's numbers are committed (tests/scale_inputs.py regenerates the inputs anywhere).
# Rows are independent (is_reachable starts from the pose; discrete mode does not carry state, C:409-462), so they are
# spread over worker processes; the result does not depend on how.
# ----------------------------------------------------------------------------------------
_G14 = {}


def _g14_solver(arm):
    if ("s", arm) not in _G14:
        _G14[("s", arm)] = make_solver(arm, 0.03)
    return _G14[("s", arm)]


def _g14_ctrl():
    if "c" not in _G14:
        _G14["c"] = quiet(ControlIK, urdf_path="../config_files/reachy2.urdf", is_dvt=False)
    return _G14["c"]


def _g14_rows_config2(job):
    arm, pos, eul, first = job
    solver = _g14_solver(arm)
    reach = np.zeros(len(pos), dtype=np.uint8)
    state = np.zeros(len(pos), dtype=np.uint8)
    sub = []
    for k, (p, e) in enumerate(zip(pos, eul)):
        if (first + k) % SCALE.SUBSAMPLE == 0:
            r = solve_symbolic(solver, p, e)  # fresh is_reachable, then get_joints(interval[0]) (Q1)
            reach[k], state[k] = r["reachable"], r["state"]
            sub.append((r["interval"], r["joints"]))
        else:
            ok, _, _, st = solver.is_reachable(np.array([p, e]))
            reach[k], state[k] = bool(ok), STATE_CODES[st]
    return reach, state, sub


def _g14_filter_config3(job):
    pos, eul = job
    solver = _g14_ctrl().symbolic_ik_solver["r_arm"]
    return np.array([bool(solver.is_reachable(np.array([p, e]))[0]) for p, e in zip(pos, eul)])


def _g14_rows_config3(Ms):
    ctrl = _g14_ctrl()
    res = [control_call(ctrl, "r_arm", M, 64, "unconstrained") for M in Ms]
    return (np.array([r[0] for r in res]), np.array([r[1] for r in res], dtype=np.uint8), np.array([r[2] for r in res], dtype=np.uint8))


def gen_scale(out, workers=None):
    import multiprocessing as mp

    workers = workers or max(1, (os.cpu_count() or 2) - 1)
    data = {"seed": np.int64(SCALE.SEED), "subsample": np.int64(SCALE.SUBSAMPLE)}
    with mp.get_context("fork").Pool(workers) as pool:
        # config 2's generator before its filter, both arms
        for arm in ARMS:
            pos, eul = SCALE.config2_unfiltered(arm)
            piece = 4096
            jobs = [(arm, pos[a:a + piece], eul[a:a + piece], a) for a in range(0, len(pos), piece)]
            parts = pool.map(_g14_rows_config2, jobs)
            reach = np.concatenate([p[0] for p in parts])
            state = np.concatenate([p[1] for p in parts])
            sub = [row for p in parts for row in p[2]]
            pre = f"c2_{arm}_"
            data[pre + "n"] = np.int64(len(pos))
            data[pre + "input_sha256"] = np.array(SCALE.sha256(np.concatenate([pos, eul], axis=1)))
            data[pre + "reachable_sha256"] = np.array(SCALE.sha256(reach))
            data[pre + "state_sha256"] = np.array(SCALE.sha256(state))
            data[pre + "state_counts"] = np.bincount(state, minlength=9).astype(np.int64)
            data[pre + "sub_reachable"] = reach[::SCALE.SUBSAMPLE].copy()
            data[pre + "sub_state"] = state[::SCALE.SUBSAMPLE].copy()
            data[pre + "sub_interval"] = np.array([r[0] for r in sub])
            data[pre + "sub_joints"] = np.array([r[1] for r in sub])
        # config 3: the candidates' is_reachable flags (the filter), then ControlIK discrete over the kept matrices
        kept, gen = [], SCALE.config3_candidates()
        while sum(int(k.sum()) for k in kept) < SCALE.N_CONFIG3:
            pos, eul = next(gen)
            piece = 8192
            flags = pool.map(_g14_filter_config3, [(pos[a:a + piece], eul[a:a + piece]) for a in range(0, len(pos), piece)])
            kept.append(np.concatenate(flags))
        kept_bits = np.packbits(np.concatenate(kept))
        _, _, kept_mask, Ms = SCALE.config3_from_kept(kept_bits)
        piece = 1024
        parts = pool.map(_g14_rows_config3, [Ms[a:a + piece] for a in range(0, len(Ms), piece)])
        joints = np.concatenate([p[0] for p in parts])
        reach = np.concatenate([p[1] for p in parts])
        state = np.concatenate([p[2] for p in parts])
        data["c3_n"] = np.int64(len(Ms))
        data["c3_candidates"] = np.int64(kept_mask.size)
        data["c3_kept_bits"] = kept_bits
        data["c3_input_sha256"] = np.array(SCALE.sha256(Ms))
        data["c3_reachable_sha256"] = np.array(SCALE.sha256(reach))
        data["c3_state_sha256"] = np.array(SCALE.sha256(state))
        data["c3_state_counts"] = np.bincount(state, minlength=9).astype(np.int64)
        data["c3_sub_reachable"] = reach[::SCALE.SUBSAMPLE].copy()
        data["c3_sub_state"] = state[::SCALE.SUBSAMPLE].copy()
        data["c3_sub_joints"] = joints[::SCALE.SUBSAMPLE].copy()
    np.savez_compressed(os.path.join(out, "g14_scale.npz"), **data)


# ----------------------------------------------------------------------------------------
# G15: the stages of is_reachable as the reference exposes them (public methods on explicit operands; its own harness,
# src/benchmark/ik_benchmarks.py:36-130, times them one by one): is_pose_in_robot_reach, get_wrist_position,
# get_limitation_wrist_circle, get_intersection_circle, are_circles_linked, points_of_nearest_approach,
# intersection_circle_line_3d_vd, utils.rotation_matrix_from_vector — chained the way the harness chains them.
# ----------------------------------------------------------------------------------------
def gen_stages(out, n=1500):
    from reachy2_symbolic_ik.utils import rotation_matrix_from_vector

    rng = np.random.default_rng(15)
    data = {}
    for arm in ARMS:
        solver = make_solver(arm, 0.03)
        # every outcome: half uniformly random poses, half reachable ones (where the circles really cross)
        pu, eu = random_poses(rng, arm, n // 2)
        pr, er = reachable_poses(rng, solver, arm, n - n // 2)
        pos, eul = np.concatenate([pu, pr]), np.concatenate([eu, er])
        rows = {k: [] for k in ("reach_ok", "reach_pos", "reach_state", "wrist", "lc", "ic_found", "ic", "linked_count", "linked",
                                "na_found", "na_q", "na_v", "cl_count", "cl_points", "rot")}
        for p, e in zip(pos, eul):
            pose = np.array([p, e])
            ok, new_pose, st = solver.is_pose_in_robot_reach(pose)
            rows["reach_ok"].append(np.uint8(bool(ok)))
            rows["reach_pos"].append(np.array(new_pose[0], dtype=float))
            rows["reach_state"].append(np.uint8(STATE_CODES[st]))
            w = np.array(solver.get_wrist_position(pose), dtype=float)
            rows["wrist"].append(w)
            solver.wrist_position = w
            lc = solver.get_limitation_wrist_circle(pose)
            rows["lc"].append(np.concatenate([lc[0], [lc[1]], lc[2]]))
            rows["rot"].append(np.array(rotation_matrix_from_vector(lc[2]), dtype=float).reshape(9))
            ic = solver.get_intersection_circle(pose)
            rows["ic_found"].append(np.uint8(ic is not None))
            linked, q, v, pts = np.array([]), np.array([]), np.full(3, NAN), None
            if ic is None:
                rows["ic"].append(np.full(7, NAN))
            else:
                rows["ic"].append(np.concatenate([ic[0], [ic[1]], ic[2]]))
                linked = np.array(solver.are_circles_linked(ic, lc), dtype=float)
                q, v = solver.points_of_nearest_approach(lc[0], lc[2], ic[0], ic[2])
                if len(q):
                    pts = solver.intersection_circle_line_3d_vd(lc[0], lc[1], v, q)
            rows["linked_count"].append(np.uint8(len(linked)))
            rows["linked"].append(linked if len(linked) else np.full(2, NAN))
            rows["na_found"].append(np.uint8(len(q) > 0))
            rows["na_q"].append(np.array(q, dtype=float) if len(q) else np.full(3, NAN))
            rows["na_v"].append(np.array(v, dtype=float))
            k = 0 if pts is None else len(pts)
            rows["cl_count"].append(np.uint8(k))
            flat = np.full(6, NAN)
            if k:
                flat[: 3 * k] = np.array(pts, dtype=float).reshape(-1)
            rows["cl_points"].append(flat)
        data[f"{arm}_pos"], data[f"{arm}_eul"] = pos, eul
        for k, v in rows.items():
            data[f"{arm}_{k}"] = np.array(v)
    # rotation_matrix_from_vector at and around its two colinear special cases (np.isclose's tolerances), any length
    vs = [[1, 0, 0], [-1, 0, 0], [3, 0, 0], [1, 1e-9, 0], [1, 2e-5, 0], [-1, 0, 1e-9], [-1, 3e-5, -2e-5], [0, 1, 0], [0, 0, -2], [1e-3, 1e-3, 1e-3],
          [1, 1.0000001e-8, 0], [1, 1e-8, 1e-8]] + rng.normal(size=(52, 3)).tolist()
    data["rot_vectors"] = np.array(vs, dtype=float)
    data["rot_matrices"] = np.array([np.array(rotation_matrix_from_vector(np.array(v, dtype=float)), dtype=float).reshape(9) for v in vs])
    # are_circles_linked on circles no pose produces: parallel planes (same / opposite normals, both sides), far apart, tangent-ish
    solver = make_solver("r_arm", 0.03)
    cases = []
    for trial in range(200):
        w = rng.uniform(-0.3, 0.3, size=3)
        n1 = rng.normal(size=3)
        kind = trial % 5
        n2 = n1 * (1.0 if kind == 0 else -1.0) if kind < 2 else rng.normal(size=3)
        if kind == 2:
            n2 = n1 + rng.normal(size=3) * 1e-8  # inside the parallel margin (after normalisation) or just outside
        c1 = w + rng.normal(size=3) * 0.05
        c2 = w + rng.normal(size=3) * (0.05 if kind < 4 else 0.6)
        r1, r2 = rng.uniform(0.02, 0.2, size=2)
        solver.wrist_position = w
        res = np.array(solver.are_circles_linked((c2, r2, n2), (c1, r1, n1)), dtype=float)
        cases.append((np.concatenate([w, c2, [r2], n2, c1, [r1], n1]), len(res), res if len(res) else np.full(2, NAN)))
    data["linked_cases_in"] = np.array([c[0] for c in cases])
    data["linked_cases_count"] = np.array([c[1] for c in cases], dtype=np.uint8)
    data["linked_cases_interval"] = np.array([c[2] for c in cases])
    np.savez_compressed(os.path.join(out, "g15_stages.npz"), **data)


# ----------------------------------------------------------------------------------------
# G16: config 5 through the reference itself at scale — 512 trajectories of the config-5 generator x 1000 control steps, each on
# a ControlIK of its own (fresh constructor state, fake clock: no timeout after the first call): digests of the `reachable` and
# `state` arrays, every 32nd trajectory's joints and previous_theta at every step.  Trajectories are independent: worker processes.
# ----------------------------------------------------------------------------------------
def _g16_walk(job):
    Ms = job  # [n_steps, k, 4, 4]
    n_steps, k = Ms.shape[:2]
    J = np.zeros((n_steps, k, 7)); F = np.zeros((n_steps, k), dtype=np.uint8); S = np.zeros((n_steps, k), dtype=np.uint8)
    TH = np.zeros((n_steps, k)); ES = np.zeros(k, dtype=np.uint8)
    real_time = ref_control_mod.time
    for a in range(k):
        clock = FakeClock()
        ref_control_mod.time = clock
        try:
            ctrl = quiet(ControlIK, urdf_path="../config_files/reachy2.urdf")
            for i in range(n_steps):
                clock.t += 1.0 / 120.0
                j, ok, st = quiet(ctrl.symbolic_inverse_kinematics, "r_arm", Ms[i, a], "continuous", d_theta_max=0.01)
                # (a latched ControlIK answers its emergency text, control_ik.py:205-210: RSIK_STATE_EMERGENCY = 8 in include/rsik.h)
                J[i, a] = np.array(j, dtype=float); F[i, a] = bool(ok); TH[i, a] = ctrl.previous_theta["r_arm"]
                S[i, a] = STATE_CODES[st] if st in STATE_CODES else (8 if ctrl.emergency_stop and st == ctrl.emergency_state else 255)
            ES[a] = bool(ctrl.emergency_stop)
        finally:
            ref_control_mod.time = real_time
    return J, F, S, TH, ES


def gen_scale_continuous(out, workers=None):
    import multiprocessing as mp

    workers = workers or max(1, (os.cpu_count() or 2) - 1)
    Ms = SCALE.config5_trajectories()
    n_steps, n_traj = Ms.shape[:2]
    piece = 4
    with mp.get_context("fork").Pool(workers) as pool:
        parts = pool.map(_g16_walk, [Ms[:, a:a + piece] for a in range(0, n_traj, piece)])
    J = np.concatenate([p[0] for p in parts], axis=1); F = np.concatenate([p[1] for p in parts], axis=1)
    S = np.concatenate([p[2] for p in parts], axis=1); TH = np.concatenate([p[3] for p in parts], axis=1)
    ES = np.concatenate([p[4] for p in parts])
    sub = slice(None, None, SCALE.SUBSAMPLE_TRAJ)
    data = {"n_traj": np.int64(n_traj), "n_steps": np.int64(n_steps), "input_sha256": np.array(SCALE.sha256(Ms)),
            "reachable_sha256": np.array(SCALE.sha256(F)), "state_sha256": np.array(SCALE.sha256(S)),
            "state_counts": np.bincount(S.ravel(), minlength=11).astype(np.int64), "emergency_stop": ES,
            "sub_joints": J[:, sub].copy(), "sub_previous_theta": TH[:, sub].copy(), "sub_reachable": F[:, sub].copy(), "sub_state": S[:, sub].copy(),
            "last_previous_theta": TH[-1].copy(), "last_joints": J[-1].copy()}
    np.savez_compressed(os.path.join(out, "g16_scale_continuous.npz"), **data)


# ----------------------------------------------------------------------------------------
# G17: the other arm, the other modes, at scale — what G14 / G16 leave out.  (a) ControlIK discrete on the l_arm mirror images of
# config 3's 256 Ki goal matrices, is_dvt=True (the singularity plane can bind: the kernels' PLANE variant), constrained_mode
# "low_elbow", 20 grid points; (b) ControlIK continuous on the l_arm mirror images of G16's first 256 trajectories x 1000 steps,
# "low_elbow", d_theta_max = 0.05.  Digests + subsamples, as there.
# ----------------------------------------------------------------------------------------
def _g17_rows_discrete(Ms):
    if "cd" not in _G14:
        _G14["cd"] = quiet(ControlIK, urdf_path="../config_files/reachy2.urdf", is_dvt=True)
    ctrl = _G14["cd"]
    res = [control_call(ctrl, "l_arm", M, 20, "low_elbow") for M in Ms]
    return (np.array([r[0] for r in res]), np.array([r[1] for r in res], dtype=np.uint8), np.array([r[2] for r in res], dtype=np.uint8))


def _g17_walk(Ms):
    n_steps, k = Ms.shape[:2]
    J = np.zeros((n_steps, k, 7)); F = np.zeros((n_steps, k), dtype=np.uint8); S = np.zeros((n_steps, k), dtype=np.uint8)
    TH = np.zeros((n_steps, k)); ES = np.zeros(k, dtype=np.uint8)
    real_time = ref_control_mod.time
    for a in range(k):
        clock = FakeClock()
        ref_control_mod.time = clock
        try:
            ctrl = quiet(ControlIK, urdf_path="../config_files/reachy2.urdf")
            for i in range(n_steps):
                clock.t += 1.0 / 120.0
                j, ok, st = quiet(ctrl.symbolic_inverse_kinematics, "l_arm", Ms[i, a], "continuous", constrained_mode="low_elbow", d_theta_max=0.05)
                J[i, a] = np.array(j, dtype=float); F[i, a] = bool(ok); TH[i, a] = ctrl.previous_theta["l_arm"]
                S[i, a] = STATE_CODES[st] if st in STATE_CODES else (8 if ctrl.emergency_stop and st == ctrl.emergency_state else 255)
            ES[a] = bool(ctrl.emergency_stop)
        finally:
            ref_control_mod.time = real_time
    return J, F, S, TH, ES


def gen_scale_variants(out, workers=None):
    import multiprocessing as mp

    workers = workers or max(1, (os.cpu_count() or 2) - 1)
    g14 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "g14_scale.npz"))
    _, _, _, Mr = SCALE.config3_from_kept(g14["c3_kept_bits"])
    Ml = SCALE.mirror_matrices(Mr)
    Mt = SCALE.mirror_matrices(SCALE.config5_trajectories()[:, :256])
    data = {}
    with mp.get_context("fork").Pool(workers) as pool:
        piece = 1024
        parts = pool.map(_g17_rows_discrete, [Ml[a:a + piece] for a in range(0, len(Ml), piece)])
        joints = np.concatenate([p[0] for p in parts]); reach = np.concatenate([p[1] for p in parts]); state = np.concatenate([p[2] for p in parts])
        data.update({"d_n": np.int64(len(Ml)), "d_input_sha256": np.array(SCALE.sha256(Ml)), "d_reachable_sha256": np.array(SCALE.sha256(reach)),
                     "d_state_sha256": np.array(SCALE.sha256(state)), "d_state_counts": np.bincount(state, minlength=9).astype(np.int64),
                     "d_sub_reachable": reach[::SCALE.SUBSAMPLE].copy(), "d_sub_state": state[::SCALE.SUBSAMPLE].copy(),
                     "d_sub_joints": joints[::SCALE.SUBSAMPLE].copy()})
        piece = 4
        parts = pool.map(_g17_walk, [Mt[:, a:a + piece] for a in range(0, Mt.shape[1], piece)])
        J = np.concatenate([p[0] for p in parts], axis=1); F = np.concatenate([p[1] for p in parts], axis=1)
        S = np.concatenate([p[2] for p in parts], axis=1); TH = np.concatenate([p[3] for p in parts], axis=1); ES = np.concatenate([p[4] for p in parts])
        sub = slice(None, None, SCALE.SUBSAMPLE_TRAJ)
        data.update({"n_traj": np.int64(Mt.shape[1]), "n_steps": np.int64(Mt.shape[0]), "input_sha256": np.array(SCALE.sha256(Mt)),
                     "reachable_sha256": np.array(SCALE.sha256(F)), "state_sha256": np.array(SCALE.sha256(S)),
                     "state_counts": np.bincount(S.ravel(), minlength=11).astype(np.int64), "emergency_stop": ES,
                     "sub_joints": J[:, sub].copy(), "sub_previous_theta": TH[:, sub].copy(), "sub_reachable": F[:, sub].copy(), "sub_state": S[:, sub].copy(),
                     "last_previous_theta": TH[-1].copy(), "last_joints": J[-1].copy()})
    np.savez_compressed(os.path.join(out, "g17_scale_variants.npz"), **data)


def ref_default_joints(k):
    return [[0.0, 0.2617993877991494, -0.17453292519943295, 0.0, 0.0, 0.0, 0.0],
            [0.0, -0.2617993877991494, 0.17453292519943295, 0.0, 0.0, 0.0, 0.0]][k]


# ----------------------------------------------------------------------------------------
# G18: the policy layer's helpers as the reference's utils module exposes them (utils.py:93-112, 334-396, 443-589: scalar functions
# that callers import, src/example/test_ik.py:16-21, test_go_to.py:10-13) on explicit arguments, and points_of_nearest_approach /
# intersection_circle_line_3d_vd on planes that are NEARLY parallel (normals 1e-6 ... 1e-3 apart: just outside
# normal_vector_margin, where a solve through the normal equations loses digits the reference's SVD keeps).
# ----------------------------------------------------------------------------------------
def gen_utils(out):
    import reachy2_symbolic_ik.utils as U

    rng = np.random.default_rng(18)
    data = {}
    pi = np.pi
    # angle_diff
    a = np.concatenate([rng.uniform(-50, 50, 300), np.array([0, pi, -pi, 2 * pi, 3 * pi, -3 * pi, 1e-17, pi / 2, 7 * pi, -7 * pi]), rng.uniform(-pi, pi, 90)])
    b = np.concatenate([rng.uniform(-50, 50, 300), np.array([0, 0, 0, 0, pi, -pi, 0, -pi / 2, 0, pi]), rng.uniform(-pi, pi, 90)])
    data["ad_a"], data["ad_b"] = a, b
    data["ad_out"] = np.array([U.angle_diff(float(x), float(y)) for x, y in zip(a, b)])
    # is_valid_angle / limit_theta_to_interval: ControlIK's intervals, the whole circle written three ways, random ones
    fixed = [[3 * pi / 4, -2 * pi / 6], [-3 * pi / 4, 2 * pi / 6], [-4 * pi / 6, 0.0], [0.0, 4 * pi / 6], [-pi, pi], [0.0, 2 * pi], [pi, -pi], [1.0, 1.0],
             [-0.5, 0.5], [0.5, -0.5], [2.5, 7.0], [-7.0, -2.5]]
    ivs = np.array(fixed * 25 + rng.uniform(-pi, pi, size=(200, 2)).tolist())
    ang = np.concatenate([rng.uniform(-10, 10, len(ivs) - 40), np.resize(np.array([pi, -pi, 0.0, 3 * pi / 4, -2 * pi / 6, 2 * pi, 1.0, 0.5]), 40)])
    prev = rng.uniform(-10, 10, len(ivs))
    data["iv_interval"], data["iv_angle"], data["iv_prev"] = ivs, ang, prev
    data["iv_valid"] = np.array([U.is_valid_angle(float(t), iv) for t, iv in zip(ang, ivs)], dtype=np.uint8)
    lim = [U.limit_theta_to_interval(float(t), float(p), iv) for t, p, iv in zip(ang, prev, ivs)]
    data["lt_theta"] = np.array([float(r[0]) for r in lim])
    data["lt_inside"] = np.array([r[1] == "theta in interval" for r in lim], dtype=np.uint8)
    assert set(r[1] for r in lim) == {"theta in interval", "theta not in interval"}
    # is_elbow_ok around its two planes, both sides, both singularity offsets
    n = 600
    solver = {arm: make_solver(arm, 0.03) for arm in ARMS}
    el = np.stack([rng.uniform(-0.2, 0.45, n), rng.uniform(-0.45, 0.45, n), rng.uniform(-0.5, 0.3, n)], axis=1)
    el[::7, 1] = np.resize(np.array([-0.2, 0.2, -0.2 - 1e-12, 0.2 + 1e-12]), len(el[::7]))
    side = np.where(rng.uniform(size=n) < 0.5, 1, -1)
    so = np.where(rng.uniform(size=n) < 0.5, 0.03, -1.01)
    coeff = np.where(rng.uniform(size=n) < 0.8, 1.0, rng.uniform(0.5, 2.0, n))
    esp = np.array([solver["r_arm" if sd == 1 else "l_arm"].elbow_singularity_position for sd in side], dtype=float)
    data["eo_elbow"], data["eo_side"], data["eo_so"], data["eo_coeff"], data["eo_esp"] = el, side.astype(float), so, coeff, esp
    data["eo_ok"] = np.array([U.is_elbow_ok(e, int(sd), float(o), float(c), p) for e, sd, o, c, p in zip(el, side, so, coeff, esp)], dtype=np.uint8)
    assert 0.05 < data["eo_ok"].mean() < 0.95
    # allow_multiturn, multiturn_safety_check, continuity_check on wound joints
    n = 400
    newj = rng.uniform(-pi, pi, size=(n, 7))
    prevj = newj + rng.normal(size=(n, 7)) * 0.3 + 2 * pi * rng.integers(-4, 5, size=(n, 7))
    data["mt_new"], data["mt_prev"] = newj, prevj
    data["mt_out"] = np.array([U.allow_multiturn(list(x), list(y), "r_arm") for x, y in zip(newj, prevj)])
    wound = prevj * rng.uniform(0.5, 1.2, size=(n, 7))
    wound[::5, 0] = np.resize(np.array([6 * pi, -6 * pi, 6 * pi + 1e-9, -6 * pi - 1e-9]), len(wound[::5]))
    limits = np.where(rng.uniform(size=(n, 3)) < 0.7, 6 * pi, rng.uniform(2.0, 20.0, size=(n, 3)))
    res = [U.multiturn_safety_check(list(j), float(l[0]), float(l[1]), float(l[2]), "before") for j, l in zip(wound, limits)]
    data["ms_joints"], data["ms_limits"] = wound, limits
    data["ms_out"] = np.array([r[0] for r in res], dtype=float)
    data["ms_stop"] = np.array([r[1] for r in res], dtype=np.uint8)
    data["ms_text"] = np.array([r[2] for r in res])
    assert 0.05 < data["ms_stop"].mean() < 0.95
    thr = np.array([0.5, 0.5, 0.5, 0.5, 1.0, 1.0, 1.0])
    cj = prevj + rng.normal(size=(n, 7)) * np.where(rng.uniform(size=(n, 1)) < 0.5, 0.15, 0.6)
    cj[::9] += 2 * pi  # (a whole turn away is continuous: angle_diff)
    res = [U.continuity_check(np.array(j), np.array(p), list(thr), "") for j, p in zip(cj, prevj)]
    data["cc_joints"], data["cc_prev"], data["cc_max"] = cj, prevj, thr
    data["cc_out"] = np.array([r[0] for r in res], dtype=float)
    data["cc_stop"] = np.array([r[1] for r in res], dtype=np.uint8)
    data["cc_text"] = np.array([r[2] for r in res])
    assert 0.1 < data["cc_stop"].mean() < 0.9
    # limit_orbita3d_joints(_wrist): random orientations, the cone's edge, SciPy's gimbal cases of the ZYZ triple (beta = 0: roll = pitch = 0)
    n = 500
    w = rng.uniform(-pi, pi, size=(n, 3))
    w[:60] *= 0.2
    w[60:70, :2] = 0.0
    w[70:80] = 0.0
    w[70:80, 2] = rng.uniform(-pi, pi, 10)
    mx = np.where(rng.uniform(size=n) < 0.7, np.deg2rad(42.5), rng.uniform(0.1, 1.4, n))
    data["lo_joints"], data["lo_max"] = w, mx
    data["lo_out"] = np.array([U.limit_orbita3d_joints(list(j), float(m)) for j, m in zip(w, mx)])
    full = rng.uniform(-pi, pi, size=(40, 7))
    data["low_joints"] = full
    data["low_out"] = np.array([U.limit_orbita3d_joints_wrist(list(j), float(np.deg2rad(42.5))) for j in full])
    # get_best_discrete_theta on the circles real poses leave on the solver, the way ControlIK calls it (control_ik.py:424-434)
    rows = {k: [] for k in ("circle", "interval", "args", "found", "theta", "worked")}
    for arm in ARMS:
        for so_ in (0.03, -1.01):
            sv = make_solver(arm, so_)
            pos, eul = reachable_poses(rng, sv, arm, 120)
            for k, (p_, e_) in enumerate(zip(pos, eul)):
                ok, interval, fn, st = sv.is_reachable(np.array([p_, e_]))
                assert ok
                nb = (20, 64, 10, 2)[k % 4]
                pref = [-4 * pi / 6, -pi + 4 * pi / 6, float(rng.uniform(-pi, pi)), float(interval[0])][(k // 4) % 4]
                prev_t = float(rng.uniform(-pi, pi))
                found, theta, text = U.get_best_discrete_theta(prev_t, interval, sv.get_elbow_position, nb, pref, arm, sv.singularity_offset,
                                                                sv.singularity_limit_coeff, sv.elbow_singularity_position)
                c = sv.intersection_circle
                rows["circle"].append(np.concatenate([c[0], [c[1]], c[2]]))
                rows["interval"].append(np.array(interval, dtype=float))
                rows["args"].append(np.concatenate([[prev_t, nb, pref, 1.0 if arm == "r_arm" else -1.0, sv.singularity_offset, sv.singularity_limit_coeff],
                                                    np.array(sv.elbow_singularity_position, dtype=float)]))
                rows["found"].append(np.uint8(bool(found)))
                rows["theta"].append(float(theta))
                rows["worked"].append(np.uint8("preferred_theta worked!" in text))
    for k, v in rows.items():
        data["bd_" + k] = np.array(v)
    assert 0.02 < data["bd_found"].mean() and 0.05 < data["bd_worked"].mean() < 0.95 and (data["bd_found"] == 0).any()
    # nearly parallel planes
    sv = make_solver("r_arm", 0.03)
    rows = {k: [] for k in ("in", "found", "q", "v", "cl_count", "cl_points", "delta")}
    for trial in range(240):
        delta = [1e-6, 2e-6, 5e-6, 1e-5, 1e-4, 1e-3][trial % 6]
        n1 = rng.normal(size=3)
        n1 /= np.linalg.norm(n1)
        t = np.cross(n1, rng.normal(size=3))
        t /= np.linalg.norm(t)
        n2 = n1 + delta * t
        n2 /= np.linalg.norm(n2)
        if trial % 12 >= 6:
            n2 = -n2
        p1 = rng.uniform(-0.3, 0.3, size=3)
        p2 = p1 + rng.normal(size=3) * 0.05
        r1 = float(rng.uniform(0.05, 0.3))
        q, v = sv.points_of_nearest_approach(p1, n1, p2, n2)
        pts = sv.intersection_circle_line_3d_vd(p1, r1, v, q) if len(q) else None
        k = 0 if pts is None else len(pts)
        flat = np.full(6, NAN)
        if k:
            flat[: 3 * k] = np.array(pts, dtype=float).reshape(-1)
        rows["in"].append(np.concatenate([p1, n1, p2, n2, [r1]]))
        rows["found"].append(np.uint8(len(q) > 0))
        rows["q"].append(np.array(q, dtype=float) if len(q) else np.full(3, NAN))
        rows["v"].append(np.array(v, dtype=float))
        rows["cl_count"].append(np.uint8(k))
        rows["cl_points"].append(flat)
        rows["delta"].append(delta)
    for k, v in rows.items():
        data["np_" + k] = np.array(v)
    # (for planes this close the two line parameters nearly agree and the reference's intersection_point often returns [] — Q7: both
    # outcomes are in the set)
    assert 0.05 < data["np_found"].mean() < 1.0, data["np_found"].mean()
    np.savez_compressed(os.path.join(out, "g18_utils.npz"), **data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    ap.add_argument("--only", default="", help="comma-separated set names (g1,g6,...): exact names")
    ap.add_argument("--check", action="store_true",
                    help="regenerate into a temporary directory and compare, array by array and byte for byte, with the fixtures "
                         "committed under --out; exit status 1 on any difference (nothing is written to --out)")
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    if args.check:
        import tempfile

        committed = out
        tmp = tempfile.TemporaryDirectory(prefix="rsik_golden_check_")
        out = tmp.name
        # (G10 reads its made-up URDF from the fixture directory: an input written for this repo, not a product of the reference)
        import shutil

        if os.path.exists(os.path.join(committed, "custom_arm.urdf")):
            shutil.copy(os.path.join(committed, "custom_arm.urdf"), os.path.join(out, "custom_arm.urdf"))
    os.makedirs(out, exist_ok=True)
    steps = [("g0", gen_constants), ("g1", gen_catalogue), ("g2", gen_sweep), ("g3", gen_reachable),
             ("g4", gen_control), ("g5", gen_helpers), ("g6", gen_continuous),
             ("g7", gen_continuous_start), ("g8", gen_matrix_edges),
             ("g9", gen_custom_geometry), ("g10", gen_custom_urdf_control), ("g11", gen_emergency),
             ("g12", gen_continuous_modes), ("g13", gen_hostile), ("g14", gen_scale), ("g15", gen_stages), ("g16", gen_scale_continuous), ("g17", gen_scale_variants),
             ("g18", gen_utils)]
    bad = 0
    for name, fn in steps:
        if args.only and name not in args.only.split(","):  # exact names: "g1" does not select "g12"
            continue
        t0 = _time.time()
        fn(out)
        print(f"{name}: done in {_time.time() - t0:.1f}s", flush=True)
        if args.check:
            # the set's file(s) just written, against the committed ones: same arrays, same dtypes and shapes, same bytes
            # (the .npz containers themselves may differ in zip metadata: the arrays are what is pinned)
            for fname in sorted(f for f in os.listdir(out) if f.startswith(name + "_")):
                new_path, old_path = os.path.join(out, fname), os.path.join(committed, fname)
                if not os.path.exists(old_path):
                    print(f"CHECK {fname}: not committed under {committed}")
                    bad += 1
                    continue
                a, b = np.load(new_path), np.load(old_path)
                diffs = [k for k in sorted(set(a.files) | set(b.files))
                         if k not in a.files or k not in b.files or a[k].dtype != b[k].dtype or a[k].shape != b[k].shape
                         or a[k].tobytes() != b[k].tobytes()]
                print(f"CHECK {fname}: " + ("identical (%d arrays)" % len(a.files) if not diffs else "DIFFERS in " + ", ".join(diffs)))
                bad += 1 if diffs else 0
                os.remove(new_path)
    if args.check:
        tmp.cleanup()
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
