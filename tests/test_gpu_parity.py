"""GPU parity tests (run with -m gpu on an MI355X): HIP kernels, through the C ABI, against
(a) golden vectors recorded from the real reference and (b) the CPU checker (oracle/) on seeded inputs.

Bars (north star): reachability flags and state codes bit-exact; joints / intervals / elbows within
1e-6 rad (observed ~1e-12; asserted at 1e-9 so that regressions show).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-9          # asserted (north-star bar: 1e-6 rad)
NORTH_STAR_TOL = 1e-6
URDF = "config_files/reachy2_ik_minimal.urdf"


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as o

    return o


def _abi_mod():
    from reachy2_symbolic_ik_amd import _abi

    return _abi


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def make_symbolic(so):
    import contextlib
    import io

    from reachy2_symbolic_ik_amd import HipSolver, SymbolicIK

    solver = HipSolver(0)
    with contextlib.redirect_stdout(io.StringIO()):
        r = SymbolicIK("r_arm", singularity_offset=so, solver=solver)
        l = SymbolicIK("l_arm", singularity_offset=so, solver=solver)
    return solver, r, l


def soa(pos, eul, torch):
    return torch.as_tensor(np.ascontiguousarray(np.concatenate([pos.T, eul.T], axis=0))).cuda()


def to_np(res):
    return {k: v.cpu().numpy() for k, v in res.items()}


def singular_rows(j):
    return np.abs(j[:, 3]) < 1e-12


def check_symbolic(res, g, prefix, tol=TOL):
    reach = g[prefix + "reachable"]
    np.testing.assert_array_equal(res["reachable"], reach)
    np.testing.assert_array_equal(res["state"], g[prefix + "state"])
    m = reach.astype(bool)
    assert np.all(np.isnan(res["joints"][~m])) and np.all(np.isnan(res["interval"][~m]))
    sing = singular_rows(g[prefix + "joints"]) & m
    for k in ("interval", "joints", "elbow"):
        mm = m & ~sing if k == "joints" else m
        err = np.max(np.abs(res[k][mm] - g[prefix + k][mm])) if mm.any() else 0.0
        assert err < tol, f"{prefix}{k}: max err {err}"
    if sing.any():  # fully extended arm: only j2 + j6 is defined (see tests/test_oracle_golden.py)
        a, b = res["joints"][sing], g[prefix + "joints"][sing]
        assert np.max(np.abs(a[:, [0, 1, 3, 4, 5]] - b[:, [0, 1, 3, 4, 5]])) < tol
        dsum = (a[:, 2] + a[:, 6]) - (b[:, 2] + b[:, 6])  # defined modulo 2 pi only (elbow yaw is not wrapped, Q3)
        assert np.max(np.abs(dsum - 2 * np.pi * np.round(dsum / (2 * np.pi)))) < 1e-7


# ------------------------------------------------------------------------------------------ rsik_solve
@pytest.mark.parametrize("tag,so", [("so003_", 0.03), ("so101_", -1.01)])
def test_catalogue_mixed_arms(golden_dir, torch_mod, tag, so):
    """G1 known-answer catalogue (reference unit-test / README / benchmark / go_to poses), r and l mixed in one
    launch through the per-pose arm byte."""
    from reachy2_symbolic_ik_amd import _abi

    g = load(golden_dir, "g1_catalogue.npz")
    solver, r, l = make_symbolic(so)
    res = to_np(solver.solve(soa(g["pos"], g["eul"], torch_mod), arm=torch_mod.as_tensor(g["arm"]).cuda(),
                             theta_policy=_abi.THETA_INTERVAL0))
    check_symbolic(res, g, tag)


def test_reference_unit_test_poses_scalar_api(golden_dir, torch_mod, capsys):
    """The reference's own tests/test_ik.py:12-79, run against the drop-in class."""
    from reachy2_symbolic_ik_amd import SymbolicIK

    symbolic_ik = SymbolicIK()
    assert "Using default parameters" in capsys.readouterr().out
    result = symbolic_ik.is_reachable([[0.4, 0.2, 0.1], [np.radians(-60), np.radians(-90), np.radians(20)]])
    assert not result[0] and len(result[1]) == 0 and result[2] is None
    result = symbolic_ik.is_reachable([[0.3, -0.2, -0.3], [0.0, np.radians(-90), 0.0]])
    assert result[0] and result[1][0] >= -np.pi and result[1][1] <= np.pi and result[2] is not None
    joints, elbow_position = result[2](result[1][0])
    assert len(joints) == 7
    result = symbolic_ik.is_reachable([[0.02, -0.2, -0.65], [0.0, 0.0, 0.0]])
    assert result[0] and np.all(result[1] == [-np.pi, np.pi]) and result[2] is not None
    joints, elbow_position = result[2](0)
    assert len(joints) == 7
    assert not symbolic_ik.is_reachable([[0.0, -0.2, -0.65], [0.0, 0.0, 0.0]])[0]
    assert not symbolic_ik.is_reachable([[0.87, -0.2, -0.0], [0.0, -np.pi / 2, 0.0]])[0]
    assert symbolic_ik.is_reachable([[0.35, -0.2, -0.28], [0.0, -np.pi / 2, 0.0]])[0]
    # README example with its recorded values
    ok, interval, fn, state = symbolic_ik.is_reachable(np.array([[0.55, -0.3, -0.15], [0, -np.pi / 2, 0]]))
    assert ok and state == "reachable"
    np.testing.assert_allclose(interval, [2.189523775249914, -0.223936328755258], atol=1e-12)
    joints, elbow = fn(interval[0])
    np.testing.assert_allclose(joints, [-1.52495747419263, -0.684394520135058, -3.931173030050867, -1.048669758375133,
                                        -0.44045940371327, 0.617944472775118, -2.281740790978251], atol=1e-11)
    assert len(elbow) in (3, 4)
    with pytest.raises(ValueError, match="arm should be either"):
        SymbolicIK(arm="middle_arm")


def test_random_sweep_all_outcomes(golden_dir, torch_mod):
    """G2: 20 000 uniform poses per arm, every outcome class (backward / out of reach / wrist ...)."""
    g = load(golden_dir, "g2_sweep.npz")
    solver, r, l = make_symbolic(0.03)
    for ik, arm in ((r, "r_arm"), (l, "l_arm")):
        res = to_np(ik.solve_batch(soa(g[f"{arm}_pos"], g[f"{arm}_eul"], torch_mod)))
        check_symbolic(res, g, f"{arm}_")


@pytest.mark.parametrize("tag,so", [("so003", 0.03), ("so101", -1.01)])
def test_reachable_theta_policies(golden_dir, torch_mod, tag, so):
    """G3: reachable poses, theta = interval[0], a fraction inside the interval, and the same theta given explicitly;
    singularity_offset 0.03 exercises the elbow-projection branch (~35 % of poses), -1.01 never does."""
    g = load(golden_dir, "g3_reachable.npz")
    solver, r, l = make_symbolic(so)
    for ik, arm in ((r, "r_arm"), (l, "l_arm")):
        p = soa(g[f"{arm}_pos"], g[f"{arm}_eul"], torch_mod)
        check_symbolic(to_np(ik.solve_batch(p)), g, f"{arm}_{tag}_i0_")
        tu = torch_mod.as_tensor(g[f"{arm}_theta_u"]).cuda()
        check_symbolic(to_np(ik.solve_batch(p, theta=("fraction", tu))), g, f"{arm}_{tag}_in_")
        th = torch_mod.as_tensor(np.nan_to_num(g[f"{arm}_{tag}_in_theta"])).cuda()
        check_symbolic(to_np(ik.solve_batch(p, theta=("explicit", th))), g, f"{arm}_{tag}_in_")
        none = to_np(ik.is_reachable_batch(p))
        np.testing.assert_array_equal(none["reachable"], g[f"{arm}_{tag}_i0_reachable"])
        assert "joints" not in none


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 1000])
def test_ragged_sizes_match_checker(torch_mod, orc, n):
    """Tail handling of the 256-thread blocks / 64-lane row transposes: any n, outputs must not bleed."""
    rng = np.random.default_rng(100 + n)
    pos = np.array([0.25, -0.2, -0.15]) + rng.uniform(-0.25, 0.25, size=(n, 3))
    eul = np.array([0, -np.pi / 2, 0]) + rng.uniform(-0.8, 0.8, size=(n, 3))
    solver, r, l = make_symbolic(0.03)
    import torch

    guard = 3
    joints = torch.full((n + guard, 7), 777.0, dtype=torch.float64, device="cuda")
    elbow = torch.full((n + guard, 3), 777.0, dtype=torch.float64, device="cuda")
    out = {"joints": joints[:n], "elbow": elbow[:n]}
    res = to_np(r.solve_batch(soa(pos, eul, torch_mod), out=out))
    ref = orc.solve_batch(orc.Arm("r_arm", 0.03), orc.Arm("l_arm", 0.03), pos, eul)
    np.testing.assert_array_equal(res["reachable"], ref["reachable"])
    np.testing.assert_array_equal(res["state"], ref["state"])
    m = ref["reachable"].astype(bool)
    assert m.sum() > 0 or n < 3
    for k in ("joints", "interval", "elbow"):
        if m.any():
            assert np.max(np.abs(res[k][m] - ref[k][m])) < TOL
    assert torch.all(joints[n:] == 777.0) and torch.all(elbow[n:] == 777.0), "store ran past the end of the batch"


def test_reach_boundary_ulps(torch_mod, orc):
    """Goal positions whose distance to the shoulder sits within a few ulps of max_arm_length (the squared-threshold
    test of the kernels, RSIK_C_MAX_LEN_SQ) and of the backward limit: states must equal the checker's bit for bit."""
    solver, r, l = make_symbolic(0.03)
    arm = orc.Arm("r_arm", 0.03)
    from reachy2_symbolic_ik_amd import constants as K

    c = K.ArmGeometry("r_arm", K.default_ik_parameters(), singularity_offset=0.03).pack()
    s, L = c[K.C_SHOULDER:K.C_SHOULDER + 3], c[K.C_MAX_LEN]
    rng = np.random.default_rng(42)
    dirs = rng.normal(size=(4000, 3))
    dirs[:, 0] = np.abs(dirs[:, 0]) + 0.2
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    scale = L * (1.0 + rng.integers(-6, 7, size=(4000, 1)) * 2.0 ** -52)
    pos = s + dirs * scale
    eul = rng.uniform(-np.pi, np.pi, size=(4000, 3))
    res = to_np(r.solve_batch(soa(pos, eul, torch_mod)))
    ref = orc.solve_batch(arm, orc.Arm("l_arm", 0.03), pos, eul)
    np.testing.assert_array_equal(res["state"], ref["state"])
    np.testing.assert_array_equal(res["reachable"], ref["reachable"])
    out = ref["state"] == 1  # RSIK_STATE_POSE_OUT_OF_REACH
    assert 0.2 < out.mean() < 0.8, "the sample must straddle the boundary"


def test_empty_batch(torch_mod):
    solver, r, l = make_symbolic(0.03)
    res = r.solve_batch(torch_mod.zeros((6, 0), dtype=torch_mod.float64, device="cuda"))
    assert res["joints"].shape == (0, 7) and res["reachable"].shape == (0,)


def test_config2_full_size_against_checker(torch_mod, orc):
    """BASELINE config 2 at full size: 1 048 576 reachable r_arm poses, theta = interval[0], checked pose by pose
    against the CPU checker (OpenMP), plus the r<->l mirror property on the same batch."""
    from bench import make_config2_poses

    pos, eul = make_config2_poses(1 << 20, seed=20250204)
    solver, r, l = make_symbolic(0.03)
    res = to_np(r.solve_batch(soa(pos, eul, torch_mod)))
    ref = orc.solve_batch(orc.Arm("r_arm", 0.03), orc.Arm("l_arm", 0.03), pos, eul, nthreads=max(1, os.cpu_count() or 1))
    np.testing.assert_array_equal(res["reachable"], ref["reachable"])
    np.testing.assert_array_equal(res["state"], ref["state"])
    assert res["reachable"].all()
    for k in ("joints", "interval", "elbow"):
        err = np.max(np.abs(res[k] - ref[k]))
        assert err < NORTH_STAR_TOL, (k, err)
        assert np.quantile(np.abs(res[k] - ref[k]), 0.9999) < 1e-9
    # mirror property (test_random_reachability.py:156-166,248): l pose = (x,-y,z; -roll,pitch,-yaw)
    posl = pos * np.array([1, -1, 1]); eull = eul * np.array([-1, 1, -1])
    resl = to_np(l.solve_batch(soa(posl, eull, torch_mod)))
    np.testing.assert_array_equal(resl["reachable"], res["reachable"])
    # theta_l = -pi - theta_r differs from interval[0] of the mirrored interval, so compare through explicit thetas
    th_r = res["interval"][:, 0]
    th_l = torch_mod.as_tensor(-np.pi - th_r).cuda()
    resl = to_np(l.solve_batch(soa(posl, eull, torch_mod), theta=("explicit", th_l)))
    sign = np.array([1, -1, -1, 1, -1, 1, -1])
    d = np.abs(resl["joints"] - res["joints"] * sign)
    d = np.minimum(d, np.abs(d - 2 * np.pi))  # wrist roll wraps at +-pi
    assert np.quantile(d, 0.9999) < 1e-9 and np.max(d) < 1e-5


# ------------------------------------------------------------------------------------------ solver-state (scalar API)
def test_scalar_state_semantics_Q1(golden_dir, torch_mod, orc):
    """get_joints is stateful like the reference's bound method: a second call after an elbow projection starts from
    the moved goal pose.  Checked call-for-call against the CPU checker's solver object."""
    g = load(golden_dir, "g3_reachable.npz")
    solver, r, l = make_symbolic(0.03)
    proj = np.where(g["r_arm_so003_i0_elbow_len"] == 3)[0][:8]
    plain = np.where(g["r_arm_so003_i0_elbow_len"] == 4)[0][:8]
    sv = orc.Solver(orc.Arm("r_arm", 0.03))
    for i in list(proj) + list(plain):
        pos, eul = g["r_arm_pos"][i], g["r_arm_eul"][i]
        ok, interval, fn, state = r.is_reachable(np.array([pos, eul]))
        oko, itvo, _ = sv.is_reachable(pos, eul)
        assert ok and oko and np.max(np.abs(interval - itvo)) < TOL
        np.testing.assert_allclose(r.wrist_position, sv.buf[6:9], atol=1e-12)
        np.testing.assert_allclose(r.intersection_circle[0], sv.buf[9:12], atol=1e-12)
        for rep in range(3):
            j, e = fn(interval[0])
            jo, eo, pr = sv.get_joints(interval[0])
            assert np.max(np.abs(j - jo)) < TOL, (i, rep)
            assert len(e) == (3 if pr else 4) and np.max(np.abs(e[:3] - eo)) < TOL
            np.testing.assert_allclose(r.goal_pose[0], sv.buf[0:3], atol=1e-12)
        e4 = r.get_elbow_position(0.3)
        assert len(e4) == 4 and np.max(np.abs(e4[:3] - sv.get_elbow_position(0.3))) < TOL


def test_helpers_elbow_and_no_limits(golden_dir, torch_mod):
    g = load(golden_dir, "g5_helpers.npz")
    solver, r, l = make_symbolic(0.03)
    for ik, arm in ((r, "r_arm"), (l, "l_arm")):
        pos, eul, th = g[f"{arm}_pos"], g[f"{arm}_eul"], g[f"{arm}_thetas"]
        for i in list(range(0, 60)) + list(range(1024, 1084)):
            ok, _, fn, _ = ik.is_reachable(np.array([pos[i], eul[i]]))
            exp = g[f"{arm}_elbow_at_theta"][i]
            assert ok == (not np.isnan(exp[0, 0]))
            if ok:
                for k in range(4):
                    assert np.max(np.abs(ik.get_elbow_position(th[i, k])[:3] - exp[k])) < TOL
            ok2, itv, fn2 = ik.is_reachable_no_limits(np.array([pos[i], eul[i]]))
            assert ok2 == bool(g[f"{arm}_nolimits_ok"][i]) and np.all(itv == [-np.pi, np.pi])
            j, e = fn2(th[i, 0])
            assert np.max(np.abs(j - g[f"{arm}_nolimits_joints"][i])) < TOL
            assert np.max(np.abs(e[:3] - g[f"{arm}_nolimits_elbow"][i])) < TOL


# ------------------------------------------------------------------------------------------ rsik_control_discrete
MODES = {"u20": (20, "unconstrained"), "u64": (64, "unconstrained"), "l20": (20, "low_elbow"), "l64": (64, "low_elbow")}


def make_control(is_dvt=False):
    import contextlib
    import io

    from reachy2_symbolic_ik_amd import ControlIK

    with contextlib.redirect_stdout(io.StringIO()):
        return ControlIK(urdf_path=URDF, is_dvt=is_dvt)


def test_control_catalogue_mixed(golden_dir, torch_mod):
    g = load(golden_dir, "g1_catalogue.npz")
    c = make_control()
    arm = torch_mod.as_tensor(g["arm"]).cuda()
    for key, (nb, mode) in MODES.items():
        c.nb_search_points = nb
        res = to_np(c.symbolic_inverse_kinematics_batch(arm, g["M"], constrained_mode=mode))
        np.testing.assert_array_equal(res["reachable"], g[f"ctrl_{key}_reachable"], err_msg=key)
        np.testing.assert_array_equal(res["state"], g[f"ctrl_{key}_state"], err_msg=key)
        err = np.max(np.abs(res["joints"] - g[f"ctrl_{key}_joints"]))
        assert err < 1e-7, (key, err)  # catalogue holds exact gimbal-lock orientations (pitch = -pi/2): Euler round trip
        assert res["emergency"].sum() == 0


def test_control_scalar_readme(torch_mod):
    """README.md:96-123 example through the drop-in ControlIK."""
    from scipy.spatial.transform import Rotation as R

    c = make_control()
    M = np.eye(4)
    M[:3, :3] = R.from_euler("xyz", [0, -np.pi / 2, 0]).as_matrix()
    M[:3, 3] = [0.55, -0.3, -0.15]
    joints, ok, state = c.symbolic_inverse_kinematics("r_arm", M, "discrete")
    assert ok and state == "reachable" and len(joints) == 7
    np.testing.assert_allclose(joints, [-0.602816657693155, -0.324538165592665, 0.008828077832103, -1.048669758375133,
                                        0.024186782799737, 0.143971244612599, -0.531677295547562], atol=1e-9)
    assert c.previous_pose["r_arm"] is M
    with pytest.raises(ValueError, match="Unknown type"):
        c.symbolic_inverse_kinematics("r_arm", M, "bogus")


@pytest.mark.parametrize("dvt_tag,is_dvt", [("std", False), ("dvt", True)])
def test_control_random(golden_dir, torch_mod, dvt_tag, is_dvt):
    """G4: random goal matrices (half uniform, half wrist-reachable), both arms, 20 / 64 search points, both
    constrained modes, DVT and non-DVT singularity offsets."""
    g = load(golden_dir, "g4_control_discrete.npz")
    c = make_control(is_dvt)
    for arm in ("r_arm", "l_arm"):
        pre = f"{dvt_tag}_{arm}_"
        M = g[pre + "M"]
        for key, (nb, mode) in MODES.items():
            if pre + key + "_joints" not in g:
                continue
            c.nb_search_points = nb
            res = to_np(c.symbolic_inverse_kinematics_batch(arm, M, constrained_mode=mode))
            np.testing.assert_array_equal(res["reachable"], g[pre + key + "_reachable"], err_msg=pre + key)
            np.testing.assert_array_equal(res["state"], g[pre + key + "_state"], err_msg=pre + key)
            err = np.max(np.abs(res["joints"] - g[pre + key + "_joints"]))
            assert err < TOL, (pre + key, err)
            assert set(np.unique(res["state"])) >= {0, 6}, "sweep hit and sweep miss must both occur"
        if dvt_tag == "std":
            c.nb_search_points = 20
            idx = g[pre + "var_idx"]
            for k in range(0, len(idx), 4):
                j, ok, st = c.symbolic_inverse_kinematics(arm, M[idx[k]], "discrete",
                                                          current_joints=list(g[pre + "var_current_joints"][k]),
                                                          preferred_theta=float(g[pre + "var_preferred_theta"][k]))
                assert ok == bool(g[pre + "var_reachable"][k])
                from reachy2_symbolic_ik_amd import STATE_STRINGS
                assert st == STATE_STRINGS[g[pre + "var_state"][k]]
                assert np.max(np.abs(np.array(j) - g[pre + "var_joints"][k])) < TOL


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("nb", [2, 3, 10, 20, 25, 33, 64, 65, 100, 200, 1000])
def test_control_grid_sizes_match_checker(golden_dir, torch_mod, orc, nb, mode):
    """Grid search strategies (rsik_set_option RSIK_OPT_SWEEP_MODE: 0 = per-wave choice, 1 = exhaustive wave-cooperative sweep, which packs
    64/pow2ceil(nb) poses per round and needs extra rounds above 64 points, 2 = per-lane search: whole grid up to 4
    points, the arc-end candidates above) must all reproduce the reference's first strict minimum."""
    g = load(golden_dir, "g4_control_discrete.npz")
    c = make_control()
    c._solver.set_option(_abi_mod().OPT_SWEEP_MODE, mode)
    c.nb_search_points = nb
    ar, al = orc.Arm("r_arm", -1.01), orc.Arm("l_arm", -1.01)
    M = np.concatenate([g["std_r_arm_M"][900:1300], g["std_l_arm_M"][900:1300]])
    arm = np.concatenate([np.zeros(400, np.uint8), np.ones(400, np.uint8)])
    res = to_np(c.symbolic_inverse_kinematics_batch(torch_mod.as_tensor(arm).cuda(), M))
    ref = orc.control_discrete_batch(ar, al, M, arm_id=arm, nb_search_points=nb)
    np.testing.assert_array_equal(res["reachable"], ref["reachable"])
    np.testing.assert_array_equal(res["state"], ref["state"])
    assert np.max(np.abs(res["joints"] - ref["joints"])) < TOL


@pytest.mark.parametrize("dvt_tag,is_dvt,offset", [("std", False, -1.01), ("dvt", True, 0.03)])
def test_control_candidate_search_any_preferred_theta(golden_dir, torch_mod, orc, dvt_tag, is_dvt, offset):
    """The per-lane candidate search (grid_theta_candidates: 4 candidates, 6 with the DVT singularity plane) under
    preferred angles other than the default: random ones, angles whose left-arm mirror image -pi - preferred_theta
    falls outside [-pi, pi] (control_ik.py:252; utils.py:468-474 then compares it unwrapped and the shortcut fails on a
    free angle: the brackets of the preferred angle join the candidates) and an angle at the edge of the range; mixed
    r / l launches, three grid sizes, per-lane search forced and per-wave choice, against the checker's exhaustive walk."""
    g = load(golden_dir, "g4_control_discrete.npz")
    c = make_control(is_dvt)
    ar, al = orc.Arm("r_arm", offset), orc.Arm("l_arm", offset)
    Mr, Ml = g[f"{dvt_tag}_r_arm_M"][::2], g[f"{dvt_tag}_l_arm_M"][1::2]   # both halves of G4: uniform and wrist-reachable goals
    M = np.concatenate([Mr, Ml])
    arm = np.concatenate([np.zeros(len(Mr), np.uint8), np.ones(len(Ml), np.uint8)])
    arm_t = torch_mod.as_tensor(arm).cuda()
    rng = np.random.default_rng(77)
    prefs = [float(x) for x in rng.uniform(-np.pi, np.pi, 4)] + [2.9, -3.1, 0.0]
    searched = 0
    for k, pref in enumerate(prefs):
        nb = (10, 20, 64)[k % 3]
        c.nb_search_points = nb
        ref = orc.control_discrete_batch(ar, al, M, arm_id=arm, nb_search_points=nb, preferred_theta=pref)
        for mode in (2, 0):
            c._solver.set_option(_abi_mod().OPT_SWEEP_MODE, mode)
            res = to_np(c.symbolic_inverse_kinematics_batch(arm_t, M, preferred_theta=pref))
            np.testing.assert_array_equal(res["reachable"], ref["reachable"], err_msg=f"pref {pref} nb {nb} mode {mode}")
            np.testing.assert_array_equal(res["state"], ref["state"], err_msg=f"pref {pref} nb {nb} mode {mode}")
            assert np.max(np.abs(res["joints"] - ref["joints"])) < TOL, (pref, nb, mode)
        searched += int(ref["reachable"].sum())
    c._solver.set_option(_abi_mod().OPT_SWEEP_MODE, 0)
    assert searched > 1000, searched


@pytest.mark.parametrize("mode", [1, 2])
def test_config3_full_size_against_checker(torch_mod, orc, mode):
    """BASELINE config 3 at full size: 262 144 wrist-reachable goal matrices, 64-point sweep, with either grid-search
    strategy forced."""
    from bench import make_config3_matrices

    M = make_config3_matrices(1 << 18, seed=20250204)
    c = make_control()
    c._solver.set_option(_abi_mod().OPT_SWEEP_MODE, mode)
    c.nb_search_points = 64
    res = to_np(c.symbolic_inverse_kinematics_batch("r_arm", M))
    ref = orc.control_discrete_batch(orc.Arm("r_arm", -1.01), orc.Arm("l_arm", -1.01), M, nb_search_points=64,
                                     nthreads=max(1, os.cpu_count() or 1))
    np.testing.assert_array_equal(res["reachable"], ref["reachable"])
    np.testing.assert_array_equal(res["state"], ref["state"])
    err = np.abs(res["joints"] - ref["joints"])
    assert np.max(err) < NORTH_STAR_TOL and np.quantile(err, 0.9999) < 1e-9
    frac = np.bincount(res["state"], minlength=7) / len(M)
    assert frac[0] > 0.5 and frac[6] > 0.1  # both the sweep-hit and the sweep-miss populations are large


def test_scale_digests_of_the_reference(golden_dir, torch_mod):
    """G14 — the reference ITSELF at BASELINE sizes, no checker in between: digests (SHA-256) of its `reachable` and `state`
    arrays over config 2's generator before filtering (1 Mi poses per arm, every outcome), over the 6.3 M candidates config 3's
    filter looked at, and over config 3's 256 Ki goal matrices through ControlIK discrete (64-point grid), plus every 64th row's
    interval and joints (oracle/gen_golden.py gen_scale; tests/scale_inputs.py regenerates the inputs from their seeds and proves
    it by their digest).  Flags and state codes bit-exact at 8.6 M poses, joints <= 1e-9 on the subsample."""
    from tests import scale_inputs as SC
    from tests.test_oracle_golden import _check_scale_set

    g = load(golden_dir, "g14_scale.npz")
    _, r, l = make_symbolic(0.03)
    for arm, ik in (("r_arm", r), ("l_arm", l)):
        pos, eul = SC.config2_unfiltered(arm)
        assert SC.sha256(np.concatenate([pos, eul], axis=1)) == str(g[f"c2_{arm}_input_sha256"]), "the seeded inputs did not regenerate"
        res = to_np(ik.solve_batch(soa(pos, eul, torch_mod)))
        _check_scale_set(g, f"c2_{arm}_", res, SC.N_CONFIG2)
    pos, eul, kept, M = SC.config3_from_kept(g["c3_kept_bits"])
    assert kept.size == int(g["c3_candidates"]) and SC.sha256(M) == str(g["c3_input_sha256"]), "the seeded inputs did not regenerate"
    c = make_control()
    flt = to_np(c.symbolic_ik_solver["r_arm"].is_reachable_batch(soa(pos, eul, torch_mod)))
    np.testing.assert_array_equal(flt["reachable"].astype(bool), kept)
    c.nb_search_points = 64
    res = to_np(c.symbolic_inverse_kinematics_batch("r_arm", M))
    _check_scale_set(g, "c3_", res, SC.N_CONFIG3)


def test_config4_full_size_mixed_arms_against_checker(torch_mod, orc):
    """BASELINE config 4 on one GPU: 1 048 576 poses with a per-pose arm byte (l poses = mirrored r poses),
    the mixed-launch kernel against the CPU checker pose by pose."""
    from bench import make_config2_poses
    from reachy2_symbolic_ik_amd import DualArmIK

    n = 1 << 20
    pos, eul = make_config2_poses(n, seed=20250204)
    arm_id = (np.random.default_rng(99).uniform(size=n) < 0.5).astype(np.uint8)
    sgn = np.where(arm_id == 1, -1.0, 1.0)
    pos = pos * np.stack([np.ones(n), sgn, np.ones(n)], axis=1)
    eul = eul * np.stack([sgn, np.ones(n), sgn], axis=1)
    dual = DualArmIK()
    res = to_np(dual.solve_batch(torch_mod.as_tensor(arm_id).cuda(), soa(pos, eul, torch_mod)))
    ref = orc.solve_batch(orc.Arm("r_arm", 0.03), orc.Arm("l_arm", 0.03), pos, eul, arm_id=arm_id,
                          nthreads=max(1, os.cpu_count() or 1))
    np.testing.assert_array_equal(res["reachable"], ref["reachable"])
    np.testing.assert_array_equal(res["state"], ref["state"])
    assert res["reachable"].all() and 0.45 < arm_id.mean() < 0.55
    for k in ("joints", "interval", "elbow"):
        err = np.abs(res[k] - ref[k])
        assert np.max(err) < NORTH_STAR_TOL and np.quantile(err, 0.9999) < 1e-9, k


# ------------------------------------------------------------------------------------------ rsik_stage (G15)
def test_stage_entry_points_against_reference(golden_dir, torch_mod):
    """rsik_stage: the stages of is_reachable on explicit operands, batched, against what the reference's own public methods returned
    for the same operands chained the way its harness chains them (G15; src/benchmark/ik_benchmarks.py:36-130): outcomes (found / not,
    how many points, empty or not) exact, numbers to 1e-9."""
    A = _abi_mod()
    g = load(golden_dir, "g15_stages.npz")
    hs, r, l = make_symbolic(0.03)
    T = lambda a: torch_mod.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).cuda()  # noqa: E731

    def close(got, want, what, tol=1e-9):
        m = ~np.isnan(want)
        assert np.array_equal(np.isnan(got), ~m), what
        assert np.max(np.abs(got[m] - want[m]), initial=0.0) < tol, (what, float(np.max(np.abs(got[m] - want[m]))))

    for arm_id, (arm, ik) in enumerate((("r_arm", r), ("l_arm", l))):
        ik._upload()
        G = lambda k: g[f"{arm}_{k}"]  # noqa: E731
        pose = np.concatenate([G("pos"), G("eul")], axis=1)
        o = hs.stage(A.STAGE_POSE_IN_REACH, T(pose), arm_id).cpu().numpy()
        np.testing.assert_array_equal(o[:, 0] != 0, G("reach_ok") != 0)
        # (the reference's "" — nothing wrong with the pose — is RSIK_STATE_EMPTY; the golden set codes it 7 as well)
        np.testing.assert_array_equal(o[:, 4].astype(np.uint8), G("reach_state"))
        close(o[:, 1:4], G("reach_pos"), "reach_pos")
        w = hs.stage(A.STAGE_WRIST_POSITION, T(pose), arm_id).cpu().numpy()
        close(w, G("wrist"), "wrist")
        lc = hs.stage(A.STAGE_LIMITATION_CIRCLE, T(np.concatenate([G("wrist"), G("pos")], axis=1)), arm_id).cpu().numpy()
        close(lc, G("lc"), "limitation circle")
        ic = hs.stage(A.STAGE_INTERSECTION_CIRCLE, T(G("wrist")), arm_id).cpu().numpy()
        np.testing.assert_array_equal(ic[:, 0] != 0, G("ic_found") != 0)
        close(ic[:, 1:], G("ic"), "intersection circle")
        have = G("ic_found") != 0
        assert 0.3 < have.mean() < 1.0
        rows = np.concatenate([G("wrist"), G("ic"), G("lc")], axis=1)[have]
        lk = hs.stage(A.STAGE_CIRCLES_LINKED, T(rows), arm_id).cpu().numpy()
        np.testing.assert_array_equal(lk[:, 0].astype(np.uint8), G("linked_count")[have])
        close(lk[:, 1:], G("linked")[have], "interval")
        assert (G("linked_count")[have] == 2).mean() > 0.3 and (G("linked_count")[have] == 0).mean() > 0.05
        lcg, icg = G("lc")[have], G("ic")[have]
        na = hs.stage(A.STAGE_NEAREST_APPROACH, T(np.concatenate([lcg[:, 0:3], lcg[:, 4:7], icg[:, 0:3], icg[:, 4:7]], axis=1)), arm_id).cpu().numpy()
        np.testing.assert_array_equal(na[:, 0] != 0, G("na_found")[have] != 0)
        found = G("na_found")[have] != 0
        close(na[found, 1:4], G("na_q")[have][found], "q")  # (round 6: a QR solve like the reference's SVD, no normal equations — 1e-9 like the rest)
        close(na[:, 4:7], G("na_v")[have], "v")
        cl = hs.stage(A.STAGE_CIRCLE_LINE, T(np.concatenate([lcg[found, 0:4], G("na_v")[have][found], G("na_q")[have][found]], axis=1)), arm_id).cpu().numpy()
        np.testing.assert_array_equal(cl[:, 0].astype(np.uint8), G("cl_count")[have][found])
        close(cl[:, 1:], G("cl_points")[have][found], "circle-line points")
        rot = hs.stage(A.STAGE_ROTATION_FROM_VECTOR, T(G("lc")[:, 4:7]), arm_id).cpu().numpy()
        close(rot, G("rot"), "rotation")
    rot = hs.stage(A.STAGE_ROTATION_FROM_VECTOR, T(g["rot_vectors"]), 0).cpu().numpy()
    close(rot, g["rot_matrices"], "rotation, special cases", tol=1e-12)
    lk = hs.stage(A.STAGE_CIRCLES_LINKED, T(g["linked_cases_in"]), 0).cpu().numpy()
    np.testing.assert_array_equal(lk[:, 0].astype(np.uint8), g["linked_cases_count"])
    close(lk[:, 1:], g["linked_cases_interval"], "made-up circles")
    assert len(set(g["linked_cases_count"].tolist())) == 2
    with pytest.raises(Exception):
        hs.stage(A.STAGE_CIRCLES_LINKED, T(np.zeros((2, 5))), 0)  # a row of the wrong length is refused on the host


def test_reference_benchmark_harness_shape_runs_on_the_drop_in(golden_dir):
    """src/benchmark/ik_benchmarks.py:12-156 in miniature: every call the reference's own per-function harness makes, made on the
    drop-in with the harness's pose — the scalar stage methods exist, chain (each reads what the one before returned, and
    self.wrist_position where the reference does) and agree with is_reachable on the same pose."""
    from reachy2_symbolic_ik_amd import SymbolicIK
    from reachy2_symbolic_ik_amd.utils import make_homogenous_matrix_from_rotation_matrix, rotation_matrix_from_vector

    import contextlib
    import io

    with contextlib.redirect_stdout(io.StringIO()):
        ik = SymbolicIK()
    goal_pose = np.array([[0.3, -0.1, 0.1], [np.radians(20), np.radians(-50), np.radians(20)]])
    ok, interval, fn, state = ik.is_reachable(goal_pose)
    joints, elbow = ik.get_joints(interval[0])
    assert ok and state == "reachable" and joints.shape == (7,)
    in_reach, pose2, st = ik.is_pose_in_robot_reach(goal_pose)
    assert in_reach and st == "" and np.array_equal(pose2, goal_pose)
    ik.wrist_position = ik.get_wrist_position(goal_pose)
    lc = ik.get_limitation_wrist_circle(goal_pose)
    ic = ik.get_intersection_circle(goal_pose)
    assert ic is not None and np.allclose(ic[0], ik.intersection_circle[0], atol=1e-12) and abs(ic[1] - ik.intersection_circle[1]) < 1e-12
    linked = ik.are_circles_linked(ic, lc)
    assert linked.shape == (2,) and np.max(np.abs(linked - interval)) < 1e-9
    R = rotation_matrix_from_vector(lc[2])
    assert R.shape == (3, 3) and np.allclose(R @ R.T, np.eye(3), atol=1e-12) and np.allclose(R[:, 0], lc[2] / np.linalg.norm(lc[2]), atol=1e-12)
    q, v = ik.points_of_nearest_approach(lc[0], lc[2], ic[0], ic[2])
    pts = ik.intersection_circle_line_3d_vd(lc[0], lc[1], v, q)
    assert q.shape == (3,) and abs(np.linalg.norm(v) - 1) < 1e-12 and (pts is None or pts.shape in ((1, 3), (2, 3)))
    assert make_homogenous_matrix_from_rotation_matrix(np.array([0.3, -0.1, 0.1]), np.eye(3)).shape == (4, 4)
    far = np.array([[2.0, -0.1, 0.1], [0.0, 0.0, 0.0]])
    in_reach, pose2, st = ik.is_pose_in_robot_reach(far)
    assert not in_reach and st == "Pose out of reach" and abs(np.linalg.norm(pose2[0] - ik.shoulder_position) - ik.max_arm_length) < 1e-6
    ik.wrist_position = ik.get_wrist_position(far)
    assert ik.get_intersection_circle(far) is None


# ------------------------------------------------------------------------------------------ csrc/rsik_math.hpp
def test_device_math_accuracy(torch_mod):
    """The kernels' own rcp / sqrt / rsqrt / atan2 / sincos / python-modulo against the host libm (float64)."""
    from reachy2_symbolic_ik_amd import HipSolver

    hs = HipSolver(0)
    rng = np.random.default_rng(5)
    n = 1 << 18
    t = lambda x: torch_mod.as_tensor(np.ascontiguousarray(x)).cuda()  # noqa: E731
    x = np.concatenate([rng.uniform(1e-6, 10.0, n // 2), 10.0 ** rng.uniform(-12, 6, n // 2)])
    r, _ = hs.debug_math(0, t(x))
    assert np.max(np.abs(r.cpu().numpy() * x - 1.0)) < 4.5e-16
    s, s2 = hs.debug_math(1, t(x))
    np.testing.assert_array_equal(s.cpu().numpy(), np.sqrt(x))   # correctly rounded
    np.testing.assert_array_equal(s2.cpu().numpy(), np.sqrt(x))
    rs, _ = hs.debug_math(2, t(x))
    assert np.max(np.abs(rs.cpu().numpy() * np.sqrt(x) - 1.0)) < 4.5e-16
    yy = np.concatenate([rng.normal(size=n - 8), [0.0, 0.0, -0.0, 1.0, -1.0, 0.0, 1e-300, 3.0]])
    xx = np.concatenate([rng.normal(size=n - 8), [0.0, -0.0, -0.0, 0.0, 0.0, -2.0, 1.0, 3.0]])
    a, _ = hs.debug_math(3, t(yy), t(xx))
    assert np.max(np.abs(a.cpu().numpy() - np.arctan2(yy, xx))) < 4.5e-16
    # the solve path's own atan2: unit vectors, table rotation + 3-term asin (dropped x^9 term < 5.3e-16 rad)
    th = np.concatenate([rng.uniform(-np.pi, np.pi, n - 9), np.array([0.0, np.pi / 2, -np.pi / 2, np.pi, np.pi / 4, -3 * np.pi / 4,
                                                                      1e-9, np.pi - 1e-9, -np.pi + 1e-9])])
    us, uc = np.sin(th), np.cos(th)
    ua, _ = hs.debug_math(7, t(us), t(uc))
    assert np.max(np.abs(ua.cpu().numpy() - np.arctan2(us, uc))) < 1.2e-15
    ang = np.concatenate([rng.uniform(-4 * np.pi, 4 * np.pi, n // 2), rng.uniform(-1e4, 1e4, n // 2 - 4),
                          [0.0, np.pi / 2, -np.pi, np.pi]])
    sn, cs = hs.debug_math(4, t(ang))
    assert np.max(np.abs(sn.cpu().numpy() - np.sin(ang))) < 3e-16
    assert np.max(np.abs(cs.cpu().numpy() - np.cos(ang))) < 3e-16
    aa = np.concatenate([rng.uniform(-30, 30, n - 6), [0.0, 2 * np.pi, -2 * np.pi, np.pi, -np.pi, 4 * np.pi]])
    bb = rng.uniform(-8, 8, n)
    m, ad = hs.debug_math(5, t(aa), t(bb))
    np.testing.assert_array_equal(m.cpu().numpy(), aa % (2 * np.pi))   # Python float modulo, bit for bit
    np.testing.assert_array_equal(ad.cpu().numpy(), ((aa - bb + np.pi) % (2 * np.pi)) - np.pi)


# ------------------------------------------------------------------------------------------ rsik_control_continuous_step
def _run_continuous(c, arm, Ms, start_joints=None, start_pose=None):
    ntraj, nsteps = Ms.shape[:2]
    st = c.new_continuous_state(arm, ntraj)
    prev = np.tile(np.asarray(c.previous_pose[arm], dtype=np.float64), (ntraj, 1, 1)) if start_pose is None else start_pose
    out = []
    for i in range(nsteps):
        res = to_np(c.symbolic_inverse_kinematics_continuous_batch(
            arm, Ms[:, i], st, timed_out=np.full(ntraj, 1 if i == 0 else 0, dtype=np.uint8), current_pose=prev,
            current_joints=(start_joints if i == 0 else None)))
        prev = Ms[:, i]
        res["theta"] = st[0].cpu().numpy().copy()
        out.append(res)
    return out, st


def test_control_continuous_golden_default_start(golden_dir, torch_mod):
    """G6: 6 trajectories x 400 steps per arm recorded from the reference with a fake clock, started from the
    constructor's default arms-along-the-body configuration — the one every real ControlIK() caller starts from
    (control_ik.py:31-35, 296-325); all trajectories of an arm advance together, one kernel launch per control step,
    state carried in HBM between launches.

    The default pose is the fully extended arm: is_reachable_no_limits pulls the wrist back onto the u + f sphere
    (symbolic_ik.py:102-105) and the elbow circle has a radius of 5.3e-5 m, so the start-up ternary search
    (utils.py:302-319) compares joint sets whose elbow-yaw / wrist-yaw split is conditioned like 1e-16 / 5e-5.  Its 16
    comparisons differ by >= 3e-4 rad (recorded from the reference: iteration 14 f1 - f2 = -3.2e-4, 15: +4.5e-4), far
    above that noise: the kernels take the reference's branch at every iteration and land on its theta,
    -1.575442041685776 (r) — asserted here from step 0: flags and states exact, carried theta <= 1e-9, joints <= 1e-7
    (north-star bar 1e-6)."""
    g = load(golden_dir, "g6_control_continuous.npz")
    c = make_control()
    for arm in ("r_arm", "l_arm"):
        Ms, J, F, S, TH = g[f"{arm}_M"], g[f"{arm}_joints"], g[f"{arm}_reachable"], g[f"{arm}_state"], g[f"{arm}_previous_theta"]
        out, st = _run_continuous(c, arm, Ms)
        for i, res in enumerate(out):
            np.testing.assert_array_equal(res["reachable"], F[:, i], err_msg=f"{arm} step {i}")
            np.testing.assert_array_equal(res["state"], S[:, i], err_msg=f"{arm} step {i}")
            assert np.max(np.abs(res["theta"] - TH[:, i])) < 1e-9, (arm, i)
            assert np.max(np.abs(res["joints"] - J[:, i])) < 1e-7, (arm, i)
        assert st[9].sum().item() == 0  # no emergency stop on these trajectories
    # the theta the start-up search must find (first step: previous_theta moves d_theta_max = 0.01 towards its target)
    assert abs(abs(g["r_arm_previous_theta"][0, 0] - (-1.575442041685776)) - 0.01) < 1e-12


def test_control_continuous_golden_explicit_start(golden_dir, torch_mod):
    """G7: explicit generic (current_joints, current_pose) start, DVT offset on odd trajectories: exact parity from
    the first step on (flags, states, joints, carried theta)."""
    g = load(golden_dir, "g7_control_continuous_start.npz")
    for is_dvt in (False, True):
        c = make_control(is_dvt)
        for arm in ("r_arm", "l_arm"):
            sel = g[f"{arm}_is_dvt"].astype(bool) == is_dvt
            Ms, J, F, S, TH = (g[f"{arm}_{k}"][sel] for k in ("M", "joints", "reachable", "state", "previous_theta"))
            out, st = _run_continuous(c, arm, Ms, start_joints=g[f"{arm}_start_joints"][sel], start_pose=g[f"{arm}_start_pose"][sel])
            for i, res in enumerate(out):
                np.testing.assert_array_equal(res["reachable"], F[:, i], err_msg=f"{arm} step {i}")
                np.testing.assert_array_equal(res["state"], S[:, i], err_msg=f"{arm} step {i}")
                assert np.max(np.abs(res["theta"] - TH[:, i])) < 1e-9, (arm, i)
                assert np.max(np.abs(res["joints"] - J[:, i])) < 1e-7, (arm, i)


def test_control_continuous_scalar_api(golden_dir, torch_mod, monkeypatch):
    """The reference call shape `symbolic_inverse_kinematics(name, M, "continuous")`, with the clock patched like the
    golden generator patched the reference's."""
    import reachy2_symbolic_ik_amd.control_ik as cik

    g = load(golden_dir, "g6_control_continuous.npz")

    class Clock:
        t = 1000.0

        @staticmethod
        def time():
            return Clock.t

    monkeypatch.setattr(cik, "time", Clock)
    for arm in ("r_arm", "l_arm"):
        c = make_control()
        Ms, J, F, S = g[f"{arm}_M"][0], g[f"{arm}_joints"][0], g[f"{arm}_reachable"][0], g[f"{arm}_state"][0]
        from reachy2_symbolic_ik_amd import STATE_STRINGS
        for i in range(60):
            Clock.t += 1.0 / 120.0
            j, ok, st = c.symbolic_inverse_kinematics(arm, Ms[i], "continuous", d_theta_max=0.01)
            assert ok == bool(F[i]) and st == STATE_STRINGS[S[i]], (arm, i)
            assert abs(c.previous_theta[arm] - g[f"{arm}_previous_theta"][0, i]) < 1e-9, (arm, i)
            assert np.max(np.abs(np.asarray(j) - J[i])) < 1e-7, (arm, i)
        assert not c.emergency_stop and not c.init


def test_control_continuous_against_checker_dvt_and_emergency(torch_mod, orc):
    """Random jumpy goal sequences (DVT singularity offset, so the elbow projection and the emergency stop both fire),
    step by step against the CPU checker's state machine."""
    c = make_control(is_dvt=True)
    rng = np.random.default_rng(77)
    ntraj, nsteps = 96, 25
    for ai, arm in enumerate(("r_arm", "l_arm")):
        a = orc.Arm(arm, 0.03)
        y = -0.2 if arm == "r_arm" else 0.2
        base = np.array([0.35, y, -0.25])
        from scipy.spatial.transform import Rotation as R
        Ms = np.zeros((ntraj, nsteps, 4, 4))
        for k in range(ntraj):
            p = base + rng.uniform(-0.15, 0.15, 3)
            e = np.array([0.0, -np.pi / 2, 0.0]) + rng.uniform(-0.5, 0.5, 3)
            for i in range(nsteps):
                jump = 0.2 if (k % 4 == 0 and i == 12) else 0.004   # a few trajectories jump -> continuity emergency stop
                p = p + rng.uniform(-jump, jump, 3)
                e = e + rng.uniform(-jump, jump, 3)
                Ms[k, i] = np.eye(4)
                Ms[k, i, :3, :3] = R.from_euler("xyz", e).as_matrix()
                Ms[k, i, :3, 3] = p
        st = c.new_continuous_state(arm, ntraj)
        states = [orc.ContinuousState(c.previous_theta[arm], c.previous_sol[arm]) for _ in range(ntraj)]
        start_j = rng.uniform(-0.6, 0.6, size=(ntraj, 7))   # generic start: no tie in the start-up search
        prev = Ms[:, 0].copy()
        prev[:, :3, 3] += rng.uniform(-0.02, 0.02, size=(ntraj, 3))
        n_em = 0
        for i in range(nsteps):
            res = to_np(c.symbolic_inverse_kinematics_continuous_batch(
                arm, Ms[:, i], st, timed_out=np.full(ntraj, 1 if i == 0 else 0, dtype=np.uint8), current_pose=prev,
                current_joints=(start_j if i == 0 else None)))
            for k in range(ntraj):
                cs = states[k]
                j, ok, code = orc.control_continuous_step(a, cs, Ms[k, i], timed_out=(i == 0), preferred_theta_arg=-4 * np.pi / 6,
                                                          preferred_theta_self=c.preferred_theta[arm], constrained_mode=0,
                                                          current_joints=(start_j[k] if i == 0 else cs.previous_sol),
                                                          current_pose=prev[k])
                assert ok == bool(res["reachable"][k]) and code == res["state"][k], (arm, k, i)
                assert np.max(np.abs(j - res["joints"][k])) < 1e-7, (arm, k, i)
            prev = Ms[:, i]
        n_em = int(st[9].sum().item())
        assert n_em == sum(s.emergency_stop for s in states) and n_em > 0


def _g12_groups(g, arm):
    """Rows of G12 that share the launch-uniform arguments: (is_dvt, mode, d_theta_max, preferred_theta) -> row indices."""
    keys = np.stack([g[f"{arm}_is_dvt"].astype(float), g[f"{arm}_mode"].astype(float), g[f"{arm}_d_theta_max"],
                     g[f"{arm}_preferred_theta"]], axis=1)
    groups = {}
    for k, key in enumerate(map(tuple, keys)):
        groups.setdefault(key, []).append(k)
    return {key: np.array(rows) for key, rows in groups.items()}


def _g12_expected_state(S):
    return np.where(S == 255, 8, S).astype(np.uint8)  # 255 in the file = the emergency text = RSIK_STATE_EMERGENCY


@pytest.mark.parametrize("is_dvt", [False, True])
def test_control_continuous_every_mode_step_kernel(golden_dir, torch_mod, is_dvt):
    """G12 through rsik_control_continuous_step (one launch per control step): the reference's continuous mode for
    constrained_mode {unconstrained, low_elbow} x d_theta_max {0.01, 0.05, 0.4} x preferred_theta argument {default,
    0.5, 2.0: outside both control intervals, so limit_theta_to_interval snaps to either end} x both starts, both arms
    (control_ik.py:225-252, 350-384; utils.py:93-127, 220-264).  Flags / states exact, carried theta <= 1e-9 at EVERY
    step, joints <= 1e-7; the trajectory that trips continuity_check with d_theta_max = 0.4 latches at the same step."""
    g = load(golden_dir, "g12_control_continuous_modes.npz")
    c = make_control(is_dvt)
    names = ["unconstrained", "low_elbow"]
    for arm in ("r_arm", "l_arm"):
        for (dvt, mode, dth, pref), rows in _g12_groups(g, arm).items():
            if bool(dvt) != is_dvt:
                continue
            M12 = g[f"{arm}_M12"][rows]                                   # [n, steps, 12]
            J, F, TH, ES = (g[f"{arm}_{k}"][rows] for k in ("joints", "reachable", "previous_theta", "emergency_stop"))
            S = _g12_expected_state(g[f"{arm}_state"][rows])
            st = c.new_continuous_state(arm, len(rows))
            prev = g[f"{arm}_start_pose"][rows]
            for i in range(M12.shape[1]):
                m12 = np.ascontiguousarray(M12[:, i].T)
                res = to_np(c.symbolic_inverse_kinematics_continuous_batch(
                    arm, m12, st, timed_out=np.full(len(rows), 1 if i == 0 else 0, dtype=np.uint8),
                    current_pose=(prev if i == 0 else None), current_joints=(g[f"{arm}_start_joints"][rows] if i == 0 else None),
                    constrained_mode=names[int(mode)], d_theta_max=float(dth), preferred_theta=float(pref)))
                tag = (arm, dvt, mode, dth, pref, i)
                np.testing.assert_array_equal(res["reachable"], F[:, i], err_msg=str(tag))
                np.testing.assert_array_equal(res["state"], S[:, i], err_msg=str(tag))
                assert np.max(np.abs(st[0].cpu().numpy() - TH[:, i])) < 1e-9, tag
                assert np.max(np.abs(res["joints"] - J[:, i])) < 1e-7, tag
                np.testing.assert_array_equal(st[9].cpu().numpy() != 0.0, ES[:, i].astype(bool), err_msg=str(tag))


@pytest.mark.parametrize("run_mode", ["pipeline", "steps"])
def test_control_continuous_every_mode_trajectory_run(golden_dir, torch_mod, run_mode):
    """G12 through rsik_control_continuous_run: the four-phase trajectory pipeline (whose theta phase replaces
    limit_theta_to_interval's comparison by a host-derived threshold per interval kind, theta_snap_plan) and the
    step-per-launch form, against the vectors recorded from the reference — every mode, rate limit and preferred-theta
    argument, both arms, DVT and not.  Flags / states exact, joints <= 1e-7, carried theta at the end <= 1e-9."""
    g = load(golden_dir, "g12_control_continuous_modes.npz")
    A = _abi_mod()
    names = ["unconstrained", "low_elbow"]
    ctrl = {False: make_control(False), True: make_control(True)}
    snapped = 0
    for arm in ("r_arm", "l_arm"):
        for (dvt, mode, dth, pref), rows in _g12_groups(g, arm).items():
            c = ctrl[bool(dvt)]
            c._solver.set_option(A.OPT_CONT_RUN_MODE, {"pipeline": A.CONT_RUN_PHASED, "steps": A.CONT_RUN_STEPS}[run_mode])
            M12 = g[f"{arm}_M12"][rows]
            J, F, TH, ES = (g[f"{arm}_{k}"][rows] for k in ("joints", "reachable", "previous_theta", "emergency_stop"))
            S = _g12_expected_state(g[f"{arm}_state"][rows])
            st = c.new_continuous_state(arm, len(rows))
            m12_steps = torch_mod.as_tensor(np.ascontiguousarray(M12.transpose(1, 2, 0))).cuda()   # [steps, 12, n]
            res = to_np(c.run_continuous_trajectories(
                arm, m12_steps, st, first_step_timed_out=True, current_joints=g[f"{arm}_start_joints"][rows],
                current_pose=g[f"{arm}_start_pose"][rows], constrained_mode=names[int(mode)], d_theta_max=float(dth),
                preferred_theta=float(pref)))
            tag = (arm, dvt, mode, dth, pref)
            np.testing.assert_array_equal(res["reachable"], F.T, err_msg=str(tag))
            np.testing.assert_array_equal(res["state"], S.T, err_msg=str(tag))
            assert np.max(np.abs(res["joints"] - np.swapaxes(J, 0, 1))) < 1e-7, tag
            assert np.max(np.abs(st[0].cpu().numpy() - TH[:, -1])) < 1e-9, tag
            np.testing.assert_array_equal(st[9].cpu().numpy() != 0.0, ES[:, -1].astype(bool), err_msg=str(tag))
            snapped += 1
            c._solver.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_AUTO)
    assert snapped == 72


def test_control_continuous_deliberate_crash(golden_dir, torch_mod, monkeypatch):
    """control_ik.py:385-387 (G12): when is_reachable_no_limits fails the reference raises RuntimeError.  ControlIK's own
    solvers cannot get there (projection_margin 1e-8 keeps the pulled-back wrist inside u + f); a solver with a negative
    margin, swapped in through the public symbolic_ik_solver attribute as in the generator, does.  The scalar call raises
    the reference's exception (and the call before / after it returns the reference's joints); the batch entry points
    report RSIK_STATE_NOT_REACHABLE_NO_LIMITS with NaN joints and leave the trajectory state alone."""
    import contextlib
    import io

    import reachy2_symbolic_ik_amd.control_ik as cik
    from reachy2_symbolic_ik_amd import SymbolicIK
    from scipy.spatial.transform import Rotation as R

    g = load(golden_dir, "g12_control_continuous_modes.npz")
    A = _abi_mod()

    class Clock:
        t = 1000.0

        @staticmethod
        def time():
            return Clock.t

    monkeypatch.setattr(cik, "time", Clock)
    eul = np.array([0.0, -np.pi / 2, 0.0])

    def mat(p):
        M = np.eye(4)
        M[:3, :3] = R.from_euler("xyz", eul).as_matrix()
        M[:3, 3] = p
        return M

    for ai, arm in enumerate(("r_arm", "l_arm")):
        c = make_control()
        with contextlib.redirect_stdout(io.StringIO()):
            c.symbolic_ik_solver[arm] = SymbolicIK(arm, projection_margin=-1e-3, singularity_offset=-1.01,
                                                   wrist_limit=np.rad2deg(c.orbita3D_max_angle), solver=c._solver)
        P, raised, J = g[f"{arm}_crash_positions"], g[f"{arm}_crash_raised"], g[f"{arm}_crash_joints"]
        start = mat(P[0])
        cj = list(cik.DEFAULT_CURRENT_JOINTS[ai])
        for i, p in enumerate(P):
            Clock.t += 1.0 / 120.0
            with contextlib.redirect_stdout(io.StringIO()):
                if raised[i][0]:
                    theta_before = c.previous_theta[arm]
                    with pytest.raises(RuntimeError) as ei:
                        c.symbolic_inverse_kinematics(arm, mat(p), "continuous", current_joints=cj, current_pose=start)
                    assert str(ei.value) == str(raised[i][1]) and c.previous_theta[arm] == theta_before
                else:
                    j, ok, st = c.symbolic_inverse_kinematics(arm, mat(p), "continuous", current_joints=cj, current_pose=start)
                    assert np.max(np.abs(np.asarray(j) - J[i])) < 1e-7, (arm, i)
        # the same three goals as one trajectory through the batch entry points
        Ms = np.stack([mat(p) for p in P])[:, None]                      # [3 steps, 1 trajectory, 4, 4]
        for run_mode in (A.CONT_RUN_STEPS, A.CONT_RUN_AUTO):
            c._solver.set_option(A.OPT_CONT_RUN_MODE, run_mode)
            st = c.new_continuous_state(arm, 1)
            res = to_np(c.run_continuous_trajectories(arm, Ms, st, first_step_timed_out=True, current_joints=np.array([cj]),
                                                      current_pose=start[None]))
            assert list(res["state"][:, 0] == A.STATE_NOT_REACHABLE_NO_LIMITS) == [False, True, False]
            assert np.all(np.isnan(res["joints"][1])) and not res["reachable"][1, 0]
            assert np.max(np.abs(res["joints"][[0, 2], 0] - J[[0, 2]])) < 1e-7
        c._solver.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_AUTO)


def test_continuous_run_replayed_from_a_graph(torch_mod):
    """rsik_control_continuous_run issues its phases on four streams tied by the context's own events; the whole run can be
    captured into a hipGraph — the side streams join the capture through those events — and replayed: same bits as the
    run issued launch by launch (bench.py --config 5 times such replays).  The capture happens on a FRESH context:
    rsik_control_continuous_reserve creates the workspace, side streams and events the run would otherwise create inside
    the capture; without it the call refuses (RSIK_E_INVALID) instead of allocating while the stream is capturing."""
    from bench import make_config5_trajectories
    from reachy2_symbolic_ik_amd import _abi

    n_traj, n_steps = 700, 150
    traj = make_config5_trajectories(n_traj, n_steps, seed=4242)
    eager = make_control()
    st0 = eager.new_continuous_state("r_arm", n_traj)
    st = st0.clone()
    out = {"joints": torch_mod.empty((n_steps, n_traj, 7), dtype=torch_mod.float64, device="cuda"),
           "reachable": torch_mod.empty((n_steps, n_traj), dtype=torch_mod.uint8, device="cuda"),
           "state": torch_mod.empty((n_steps, n_traj), dtype=torch_mod.uint8, device="cuda")}

    def one(c):
        st.copy_(st0)
        c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0], out=out)

    one(eager)
    torch_mod.cuda.synchronize()
    ref = {k: v.clone() for k, v in out.items()}
    ref["cont_state"] = st.clone()

    side = torch_mod.cuda.Stream()
    # (a) a fresh context that has reserved nothing refuses the capture with a clear message (and leaves the capture usable)
    fresh = make_control()
    fresh._upload_arms()
    with torch_mod.cuda.stream(side):
        g0 = torch_mod.cuda.CUDAGraph()
        with torch_mod.cuda.graph(g0, stream=side):
            with pytest.raises(_abi.RsikError) as ei:
                one(fresh)
            assert "rsik_control_continuous_reserve" in str(ei.value)
    # (b) after reserve the same context captures the run without ever having run it
    fresh._solver.control_continuous_reserve(n_traj, n_steps)
    with torch_mod.cuda.stream(side):
        g = torch_mod.cuda.CUDAGraph()
        with torch_mod.cuda.graph(g, stream=side):  # a refused capture is the regression this test exists to catch
            one(fresh)
    for v in out.values():
        v.zero_()
    st.zero_()
    for _ in range(2):
        g.replay()
    torch_mod.cuda.synchronize()
    for k, v in out.items():
        assert torch_mod.equal(ref[k].view(torch_mod.uint8), v.view(torch_mod.uint8)), k
    assert torch_mod.equal(ref["cont_state"].view(torch_mod.uint8), st.view(torch_mod.uint8))
    # (c) a later, larger eager run on the same context outgrows the workspace: the graph must stay valid (the old
    # workspace is retired, not freed)
    big = make_config5_trajectories(n_traj * 3, n_steps * 2, seed=5)
    st_big = fresh.new_continuous_state("r_arm", n_traj * 3)
    fresh.run_continuous_trajectories("r_arm", big, st_big, first_step_timed_out=True, current_pose=big[0])
    torch_mod.cuda.synchronize()
    for v in out.values():
        v.zero_()
    g.replay()
    torch_mod.cuda.synchronize()
    for k, v in out.items():
        assert torch_mod.equal(ref[k].view(torch_mod.uint8), v.view(torch_mod.uint8)), k


def test_capture_continuous_trajectories_convenience(torch_mod):
    """ControlIK.capture_continuous_trajectories: the run recorded on a fresh context (two blocks under capture, four when
    issued eagerly: the results do not depend on the cut) and replayed gives the eager run's bits."""
    from bench import make_config5_trajectories

    n_traj, n_steps = 520, 203
    traj = make_config5_trajectories(n_traj, n_steps, seed=77)
    eager = make_control()
    st0 = eager.new_continuous_state("r_arm", n_traj)
    st = st0.clone()
    ref = eager.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0])
    torch_mod.cuda.synchronize()
    c = make_control()
    st2 = st0.clone()
    graph, out = c.capture_continuous_trajectories("r_arm", traj, st2, first_step_timed_out=True, current_pose=traj[0])
    for _ in range(2):
        st2.copy_(st0)
        graph.replay()
    torch_mod.cuda.synchronize()
    for k in ref:
        assert torch_mod.equal(ref[k].view(torch_mod.uint8), out[k].view(torch_mod.uint8)), k
    assert torch_mod.equal(st.view(torch_mod.uint8), st2.view(torch_mod.uint8))


def test_capture_continuous_trajectories_takes_host_arguments(torch_mod):
    """Per-trajectory arguments handed over as host arrays (arm ids, current_joints, current_pose as [n,4,4] numpy) are brought
    to the device before the capture begins and kept alive on the graph; a wrong `out` buffer or an unknown argument is
    refused before anything is recorded (and the context stays usable)."""
    from bench import make_config5_trajectories

    n_traj, n_steps = 130, 40
    traj = make_config5_trajectories(n_traj, n_steps, seed=5)
    arm = (np.arange(n_traj) % 2).astype(np.uint8)
    cj = np.zeros((n_traj, 7))
    first = traj[0].T.cpu().numpy()
    pose = np.tile(np.eye(4), (n_traj, 1, 1))
    pose[:, :3, :3] = first[:, :9].reshape(n_traj, 3, 3)
    pose[:, :3, 3] = first[:, 9:]
    eager = make_control()
    st0 = eager.new_continuous_state("r_arm", n_traj)
    st = st0.clone()
    ref = eager.run_continuous_trajectories(torch_mod.as_tensor(arm).cuda(), traj, st, current_joints=cj, current_pose=pose)
    torch_mod.cuda.synchronize()
    c = make_control()
    st2 = st0.clone()
    with pytest.raises(TypeError):
        c.capture_continuous_trajectories(arm, traj, st2, no_such_argument=1)
    with pytest.raises(ValueError):
        c.capture_continuous_trajectories(arm, traj, st2, out={"joints": torch_mod.empty((1,), device="cuda")})
    graph, out = c.capture_continuous_trajectories(arm, traj, st2, current_joints=cj, current_pose=pose)
    assert set(graph.rsik_inputs) == {"arm", "current_joints", "current_pose"}
    graph.replay()
    torch_mod.cuda.synchronize()
    for k in ref:
        assert torch_mod.equal(ref[k].view(torch_mod.uint8), out[k].view(torch_mod.uint8)), k
    assert torch_mod.equal(st.view(torch_mod.uint8), st2.view(torch_mod.uint8))


def test_two_threads_two_contexts(torch_mod, orc):
    """include/rsik.h: a context is used by one thread at a time; contexts are independent.  Two threads, each with a
    context and a stream of its own, solve different batches concurrently (rsik_solve and the continuous pipeline, whose
    workspace and side streams are per context): each must get exactly what it gets alone."""
    import threading

    from bench import make_config5_trajectories
    from reachy2_symbolic_ik_amd import HipSolver, SymbolicIK

    rng = np.random.default_rng(99)
    jobs = []
    for k in range(2):
        n = 50_000 + 7_000 * k
        pos = np.array([0.0, -0.2, 0.0]) + rng.uniform(-0.7, 0.7, size=(n, 3))
        eul = rng.uniform(-np.pi, np.pi, size=(n, 3))
        traj = make_config5_trajectories(300 + 50 * k, 96, seed=31 + k)
        jobs.append({"pos": pos, "eul": eul, "traj": traj})

    def work(job, reps):
        import contextlib
        import io

        stream = torch_mod.cuda.Stream()
        with torch_mod.cuda.stream(stream):
            solver = HipSolver(0)
            with contextlib.redirect_stdout(io.StringIO()):
                ik = SymbolicIK("r_arm", solver=solver)
            c = make_control()
            soa = soa_of = torch_mod.as_tensor(np.ascontiguousarray(np.concatenate([job["pos"].T, job["eul"].T], axis=0))).cuda()
            res = None
            for _ in range(reps):
                res = ik.solve_batch(soa_of)
                st = c.new_continuous_state("r_arm", job["traj"].shape[2])
                run = c.run_continuous_trajectories("r_arm", job["traj"], st, first_step_timed_out=True, current_pose=job["traj"][0])
            stream.synchronize()
            job["got"] = {k: v.clone() for k, v in res.items()}
            job["got_run"] = {k: v.clone() for k, v in run.items()}
            del soa

    for job in jobs:  # alone, one after the other
        work(job, 1)
        job["alone"], job["alone_run"] = job.pop("got"), job.pop("got_run")
    threads = [threading.Thread(target=work, args=(job, 5)) for job in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch_mod.cuda.synchronize()
    for job in jobs:
        for k, v in job["alone"].items():
            assert torch_mod.equal(v.view(torch_mod.uint8), job["got"][k].view(torch_mod.uint8)), k
        for k, v in job["alone_run"].items():
            assert torch_mod.equal(v.view(torch_mod.uint8), job["got_run"][k].view(torch_mod.uint8)), k
    ref = orc.solve_batch(orc.Arm("r_arm", 0.03), orc.Arm("l_arm", 0.03), jobs[0]["pos"][:2000], jobs[0]["eul"][:2000])
    np.testing.assert_array_equal(jobs[0]["got"]["state"][:2000].cpu().numpy(), ref["state"])


def _same_run(torch, ref, got, tag, joint_tol=1e-9):
    """Two runs of the same trajectories (pipeline vs step kernel, or two block sizes of the pipeline): flags, state codes,
    the carried theta (row 0 of the trajectory state) and its flag rows bit for bit; joints and previous_sol to
    `joint_tol`.  (The pipeline's joints phase writes a quiet step as raw joint + whole turns instead of previous +
    angle_diff(raw, previous) and rebuilds the goal rotation's third row from the other two: both within the last bit of
    their inputs, no accumulation — but where the arm is stretched out the elbow-yaw / wrist-yaw split amplifies a last bit
    2e4 times and more (see test_control_continuous_golden_default_start; 1.2e-10 over 32 M eventful trajectory-steps,
    scripts/soak_pipeline.py), hence 1e-9 and not 1e-14.)"""
    for k in ref:
        a, b = ref[k], got[k]
        if k == "joints":
            assert float((a - b).abs().max()) <= joint_tol, (tag, k, float((a - b).abs().max()))
        elif k == "cont_state":
            assert torch.equal(a[0].view(torch.uint8), b[0].view(torch.uint8)), (tag, "previous_theta")
            assert float((a[1:8] - b[1:8]).abs().max()) <= joint_tol, (tag, "previous_sol")
            assert torch.equal(a[8:11], b[8:11]), (tag, "init / emergency / has_previous_sol")
        else:
            assert torch.equal(a.view(torch.uint8), b.view(torch.uint8)), (tag, k)


@pytest.mark.parametrize("n_traj,n_steps", [(1, 300), (2, 77), (7, 129), (65, 33), (4099, 40)])
def test_continuous_pipeline_odd_batch_shapes(torch_mod, n_traj, n_steps):
    """The pipeline's sequential phases address their arrays as (buffer, row, lane) and run single-wave workgroups: a
    single trajectory (the reference's own use), a few, one more than a wave, one more than 64 waves — against the step
    kernel (flags, states and the carried theta bit for bit, joints to 1e-12: _same_run)."""
    from bench import make_config5_trajectories

    A = _abi_mod()
    traj = make_config5_trajectories(n_traj, n_steps, seed=1000 + n_traj)
    c = make_control()
    ref = None
    for run_mode in (A.CONT_RUN_STEPS, A.CONT_RUN_PHASED):
        c._solver.set_option(A.OPT_CONT_RUN_MODE, run_mode)
        st = c.new_continuous_state("r_arm", n_traj)
        res = c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0])
        torch_mod.cuda.synchronize()
        got = {k: v.clone() for k, v in res.items()}
        got["cont_state"] = st[:11].clone()
        if ref is None:
            ref = got
        else:
            _same_run(torch_mod, ref, got, (n_traj, n_steps))
    c._solver.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_AUTO)


def test_continuous_pipeline_every_short_length(torch_mod):
    """The sequential phases fetch their operands in batches (16 steps / 16 chunks) with the next batch in flight, and the
    loops that do it have a turn of two batches, a tail for two, one or no full batch left, and a partial batch — for a run's
    first block (whose first step is taken separately) and for the blocks behind it.  Every run length from 1 to 80 steps
    as one block, and every length from 49 to 130 cut into blocks of 48, against one launch of the step kernel per step."""
    from bench import make_config5_trajectories

    A = _abi_mod()
    n_traj = 72
    traj_all = make_config5_trajectories(n_traj, 130, seed=4242)
    c = make_control()
    for blk, lengths in ((0, range(1, 81)), (48, range(49, 131))):
        for n_steps in lengths:
            traj = traj_all[:n_steps].contiguous()
            ref = None
            for run_mode in (A.CONT_RUN_STEPS, A.CONT_RUN_PHASED):
                c._solver.set_option(A.OPT_CONT_RUN_MODE, run_mode)
                c._solver.set_option(A.OPT_CONT_BLOCK_STEPS, blk)
                st = c.new_continuous_state("r_arm", n_traj)
                res = c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0])
                torch_mod.cuda.synchronize()
                got = {k: v.clone() for k, v in res.items()}
                got["cont_state"] = st[:11].clone()
                if ref is None:
                    ref = got
                else:
                    _same_run(torch_mod, ref, got, (blk, n_steps))
    c._solver.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_AUTO)
    c._solver.set_option(A.OPT_CONT_BLOCK_STEPS, 0)


@pytest.mark.parametrize("arm", ["r_arm", "l_arm"])
@pytest.mark.parametrize("mode", ["unconstrained", "low_elbow"])
@pytest.mark.parametrize("d_theta_max", [0.01, 0.4])
def test_continuous_pipeline_theta_step_per_interval_kind(torch_mod, arm, mode, d_theta_max):
    """The theta phase of the trajectory pipeline runs a step specialised for the launch's control interval (wrap-around
    for the right arm unconstrained, inner intervals for the others — one of them starting at -pi), with the choice of
    the nearer interval end reduced to one threshold the host derives (theta_snap_plan, rsik_lib.hip).  The step kernel
    (one launch per control step) keeps limit_theta_to_interval's own arithmetic: both must give the same bits for every
    arm x constrained mode, also with a rate limit large enough for theta to cross the gap's middle in one step."""
    from bench import make_config5_trajectories

    A = _abi_mod()
    n_traj, n_steps = 520, 150
    traj = make_config5_trajectories(n_traj, n_steps, seed=7 + len(arm) + len(mode))
    if arm == "l_arm":  # the left arm's side of the workspace: y -> -y (rows 0-8 are the rotation, 9-11 x y z)
        traj = traj.clone()
        traj[:, 10] = -traj[:, 10]
    c = make_control()
    ref = None
    for run_mode in (A.CONT_RUN_STEPS, A.CONT_RUN_PHASED):
        c._solver.set_option(A.OPT_CONT_RUN_MODE, run_mode)
        st = c.new_continuous_state(arm, n_traj)
        res = c.run_continuous_trajectories(arm, traj, st, first_step_timed_out=True, current_pose=traj[0],
                                            constrained_mode=mode, d_theta_max=d_theta_max)
        torch_mod.cuda.synchronize()
        got = {k: v.clone() for k, v in res.items()}
        got["cont_state"] = st[:11].clone()
        if ref is None:
            ref = got
            assert bool(torch_mod.isfinite(ref["joints"]).all())
        else:
            _same_run(torch_mod, ref, got, (arm, mode, d_theta_max))
    c._solver.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_AUTO)


@pytest.mark.parametrize("run_mode", ["loop", "steps"])
def test_continuous_run_equals_stepwise(golden_dir, torch_mod, run_mode):
    """rsik_control_continuous_run against G7 and against the step-by-step API, both ways it can issue the work: one
    launch whose kernel walks all steps with the trajectory state in registers ("loop", what small batches get) and
    one launch per step ("steps", what chip-filling batches get)."""
    g = load(golden_dir, "g7_control_continuous_start.npz")
    c = make_control()
    A = _abi_mod()
    c._solver.set_option(A.OPT_CONT_RUN_MODE, {"loop": A.CONT_RUN_PHASED, "steps": A.CONT_RUN_STEPS}[run_mode])
    for arm in ("r_arm", "l_arm"):
        sel = ~g[f"{arm}_is_dvt"].astype(bool)
        Ms, J, F, S = (g[f"{arm}_{k}"][sel] for k in ("M", "joints", "reachable", "state"))
        st = c.new_continuous_state(arm, Ms.shape[0])
        res = to_np(c.run_continuous_trajectories(arm, np.swapaxes(Ms, 0, 1), st, first_step_timed_out=True,
                                                  current_joints=g[f"{arm}_start_joints"][sel],
                                                  current_pose=g[f"{arm}_start_pose"][sel]))
        np.testing.assert_array_equal(res["reachable"], F.T)
        np.testing.assert_array_equal(res["state"], S.T)
        assert np.max(np.abs(res["joints"] - np.swapaxes(J, 0, 1))) < 1e-7
        out, st2 = _run_continuous(c, arm, Ms, start_joints=g[f"{arm}_start_joints"][sel], start_pose=g[f"{arm}_start_pose"][sel])
        _same_run(torch_mod, {"cont_state": st[:11]}, {"cont_state": st2[:11]}, (arm, run_mode))
    # both arms' trajectories in ONE mixed launch (per-trajectory arm byte)
    sel = {a: ~g[f"{a}_is_dvt"].astype(bool) for a in ("r_arm", "l_arm")}
    cat = lambda k: np.concatenate([g[f"r_arm_{k}"][sel["r_arm"]], g[f"l_arm_{k}"][sel["l_arm"]]])  # noqa: E731
    Ms, J, F, S = cat("M"), cat("joints"), cat("reachable"), cat("state")
    arm_id = torch_mod.as_tensor(np.concatenate([np.zeros(sel["r_arm"].sum(), np.uint8), np.ones(sel["l_arm"].sum(), np.uint8)])).cuda()
    st = c.new_continuous_state(arm_id, Ms.shape[0])
    res = to_np(c.run_continuous_trajectories(arm_id, np.swapaxes(Ms, 0, 1), st, first_step_timed_out=True,
                                              current_joints=cat("start_joints"), current_pose=cat("start_pose")))
    np.testing.assert_array_equal(res["reachable"], F.T)
    np.testing.assert_array_equal(res["state"], S.T)
    assert np.max(np.abs(res["joints"] - np.swapaxes(J, 0, 1))) < 1e-7


def test_continuous_pipeline_block_sizes_agree(torch_mod):
    """The trajectory pipeline cuts a run into blocks (RSIK_OPT_CONT_BLOCK_STEPS) whose four phases overlap on four
    streams; the operands of the two sequential phases are fetched 16 / 32 steps at a time.  Whatever the block size —
    shorter than a batch, not a multiple of one, the whole run — results and carried state must be the same bits, and
    equal to one launch of the step kernel per control step."""
    from bench import make_config5_trajectories

    A = _abi_mod()
    n_traj, n_steps = 300, 131
    traj = make_config5_trajectories(n_traj, n_steps, seed=99)
    c = make_control()
    ref = None
    for mode, blk in ((A.CONT_RUN_STEPS, 0), (A.CONT_RUN_PHASED, 0), (A.CONT_RUN_PHASED, 5), (A.CONT_RUN_PHASED, 32),
                      (A.CONT_RUN_PHASED, 40), (A.CONT_RUN_PHASED, 131), (A.CONT_RUN_PHASED, 4000)):
        c._solver.set_option(A.OPT_CONT_RUN_MODE, mode)
        c._solver.set_option(A.OPT_CONT_BLOCK_STEPS, blk)
        st = c.new_continuous_state("r_arm", n_traj)
        res = c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0])
        torch_mod.cuda.synchronize()
        got = {k: v.clone() for k, v in res.items()}
        got["cont_state"] = st[:11].clone()
        if ref is None:
            ref = got
            assert bool(torch_mod.isfinite(ref["joints"]).all()) and 0.05 < float(ref["reachable"].float().mean()) < 0.95
        else:
            _same_run(torch_mod, ref, got, (mode, blk))
    c._solver.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_AUTO)
    c._solver.set_option(A.OPT_CONT_BLOCK_STEPS, 0)


@pytest.mark.parametrize("euler_mode", ["auto", "never"])
def test_continuous_pipeline_takes_slightly_skewed_matrices_as_they_are(torch_mod, euler_mode):
    """The pipeline's joints phase rebuilds a goal rotation's third row as row 0 x row 1 only where the prepare phase found the
    two equal to 1e-14.  A matrix whose third row is off by 1e-10 — through SciPy's nearest-rotation round trip under
    RSIK_EULER_AUTO, taken as it is under RSIK_EULER_NEVER — must go through every form with the entries it has: the
    pipeline forms and the step kernel agree to 1e-9 (where the arm is stretched out a rebuilt row would differ by more)."""
    from bench import make_config5_trajectories

    A = _abi_mod()
    n_traj, n_steps = 500, 96
    traj = make_config5_trajectories(n_traj, n_steps, seed=4242).clone()
    g = torch_mod.Generator(device="cpu").manual_seed(5)
    traj[:, 6:9, :] += (torch_mod.rand((n_steps, 3, n_traj), generator=g, dtype=torch_mod.float64).to(traj.device) - 0.5) * 2e-10
    c = make_control()
    hs = c._solver
    hs.set_option(A.OPT_EULER_ROUNDTRIP, A.EULER_NEVER if euler_mode == "never" else A.EULER_AUTO)
    ref = None
    for run_mode in (A.CONT_RUN_STEPS, A.CONT_RUN_PHASED):
        hs.set_option(A.OPT_CONT_RUN_MODE, run_mode)
        st = c.new_continuous_state("r_arm", n_traj)
        res = c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0])
        hs.synchronize()
        got = {k: v.clone() for k, v in res.items()}
        got["cont_state"] = st[:11].clone()
        if ref is None:
            ref = got
        else:
            _same_run(torch_mod, ref, got, (euler_mode, run_mode), joint_tol=1e-9)
    hs.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_AUTO)
    hs.set_option(A.OPT_EULER_ROUNDTRIP, A.EULER_AUTO)


def test_continuous_pipeline_through_gimbal_lock(torch_mod):
    """Config 5's generator swings the goal's pitch through -pi/2: within 1e-5 of the lock the control kernels make ControlIK's
    matrix -> Euler -> matrix round trip (RSIK_EULER_AUTO), which evaluates sines and cosines from the workgroup's table in LDS —
    in the pipeline's joints phase on a rarely taken branch.  At BASELINE size eight trajectory-steps in four million take it; here
    every step of every trajectory sits 0 ... 2e-5 rad from the lock (both sides, the lock itself included), so a branch
    that ran ahead of the table's staging, or derived another rotation than the prepare phase, shows: pipeline forms against the
    step kernel, flags / states / carried theta bit for bit, joints to 1e-9."""
    A = _abi_mod()
    n_traj, n_steps = 777, 160
    dev = torch_mod.device("cuda", 0)
    g = torch_mod.Generator(device="cpu").manual_seed(99)
    t = (torch_mod.arange(n_steps, dtype=torch_mod.float64)[:, None] / 120.0 + 11.0 + torch_mod.rand(n_traj, generator=g, dtype=torch_mod.float64)[None, :] * 40.0).to(dev)
    c0 = [0.65, -0.2, 0.0, 0.0, -np.pi / 2, 0.0]
    amp = [0.35, 0.35, 0.35, np.pi / 6, np.pi / 6, np.pi / 6]
    freq = [0.6, 0.34, 0.78, 0.18, 0.31, 0.47]
    v = [c + a * torch_mod.sin(f * t) for c, a, f in zip(c0, amp, freq)]
    # the pitch stays within 2e-5 rad of the lock all the way (a smooth swing: no jump for the continuity check to latch on), on
    # both sides of goal_from_m12's 1e-10 threshold on |R20|, and every eighth step sits on the lock itself
    v[4] = -np.pi / 2 + 2e-5 * torch_mod.sin(freq[4] * 40.0 * t)
    v[4][::8] = -np.pi / 2
    ca, sa, cb, sb, cc, sc = (torch_mod.cos(v[3]), torch_mod.sin(v[3]), torch_mod.cos(v[4]), torch_mod.sin(v[4]), torch_mod.cos(v[5]), torch_mod.sin(v[5]))
    rows = [cc * cb, cc * sb * sa - sc * ca, cc * sb * ca + sc * sa, sc * cb, sc * sb * sa + cc * ca, sc * sb * ca - cc * sa,
            -sb, cb * sa, cb * ca, v[0], v[1], v[2]]
    traj = torch_mod.stack(rows, dim=1).contiguous()
    frac_lock = float((traj[:, 6, :].abs() > 1.0 - 1e-10).double().mean())  # (what goal_from_m12 takes for the lock)
    assert 0.2 < frac_lock < 0.9, frac_lock
    c = make_control()
    hs = c._solver
    ref = None
    for run_mode, blk in ((A.CONT_RUN_STEPS, 0), (A.CONT_RUN_PHASED, 0), (A.CONT_RUN_PHASED, 48)):
        hs.set_option(A.OPT_CONT_RUN_MODE, run_mode)
        hs.set_option(A.OPT_CONT_BLOCK_STEPS, blk)
        st = c.new_continuous_state("r_arm", n_traj)
        res = c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0], d_theta_max=0.05)
        hs.synchronize()
        got = {k: v_.clone() for k, v_ in res.items()}
        got["cont_state"] = st[:11].clone()
        if ref is None:
            ref = got
        else:
            _same_run(torch_mod, ref, got, ("gimbal lock", run_mode, blk), joint_tol=1e-9)
    hs.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_AUTO)
    hs.set_option(A.OPT_CONT_BLOCK_STEPS, 0)


def _eventful_trajectories(torch, n_traj, n_steps, seed, arm):
    """[n_steps, 12, n_traj] goal matrices: config 5's sinusoids plus, per trajectory, one of: a jump of the goal, a winding
    wrist (reaches the +-6 pi limit), a stretch far out of reach, a run of exact repeats ("stay" steps) — what makes the
    sequential phases take their rare paths (scripts/soak_pipeline.py's generator, smaller)."""
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *sh: torch.rand(*sh, generator=g, dtype=torch.float64).to(dev)  # noqa: E731
    k = torch.arange(n_steps, dtype=torch.float64, device=dev)[:, None]
    t = k / 60.0 + 11.0 + r(n_traj)[None, :] * 40.0
    c0 = [0.45, -0.2 if arm == "r_arm" else 0.2, -0.1, 0.0, -np.pi / 2, 0.0]
    amp = [0.3, 0.3, 0.3, np.pi / 4, np.pi / 4, np.pi / 4]
    freq = [0.6, 0.34, 0.78, 0.18, 0.31, 0.47]
    v = [c + a * torch.sin(f * t) for c, a, f in zip(c0, amp, freq)]
    kind = (r(n_traj) * 6).floor()
    at = (r(n_traj) * max(n_steps - 20, 1) + 5).floor()
    after = (k >= at[None, :]).double()
    v[0] = v[0] + after * (kind < 2).double()[None, :] * 0.25 * (r(n_traj)[None, :] - 0.5)
    v[5] = v[5] + (kind == 2).double()[None, :] * k * 0.12 + (kind == 3).double()[None, :] * k * -0.2
    far = ((k >= at[None, :]) & (k < at[None, :] + 25)).double() * (kind == 4).double()[None, :]
    v[0] = v[0] + far * 2.0
    hold = ((k >= at[None, :]) & (k < at[None, :] + 6)).double() * (kind == 5).double()[None, :]
    idx = torch.where(hold.bool(), at[None, :].expand(n_steps, -1), k.expand(-1, n_traj)).long().clamp(max=n_steps - 1)
    v = [torch.gather(x, 0, idx) for x in v]
    ca, sa, cb, sb, cc, sc = torch.cos(v[3]), torch.sin(v[3]), torch.cos(v[4]), torch.sin(v[4]), torch.cos(v[5]), torch.sin(v[5])
    rows = [cc * cb, cc * sb * sa - sc * ca, cc * sb * ca + sc * sa, sc * cb, sc * sb * sa + cc * ca, sc * sb * ca - cc * sa,
            -sb, cb * sa, cb * ca, v[0], v[1], v[2]]
    return torch.stack(rows, dim=1).contiguous()


@pytest.mark.parametrize("n_traj,n_steps,arm,mode,dmax", [(1, 300, "r_arm", "unconstrained", 0.01), (7, 129, "l_arm", "low_elbow", 0.3),
                                                         (65, 33, "r_arm", "low_elbow", 0.01), (600, 203, "l_arm", "unconstrained", 0.3),
                                                         (4099, 40, "r_arm", "unconstrained", 0.01)])
def test_continuous_run_forms_are_bit_identical(torch_mod, n_traj, n_steps, arm, mode, dmax):
    """rsik_control_continuous_run has several forms of issuing the same four bodies (include/rsik.h: the phased pipeline and
    its variants — events instead of stream value words, no theta-first hold, other block sizes).  They run the same device
    code on the same data: every output and
    the carried state must be the same BITS, on eventful trajectories (jumps, wound wrists, unreachable stretches, repeats)
    that take the sequential phases through their rare paths; the step kernel, launch per step, bounds them all (_same_run).
    One exception since round 6: a cut into MORE than eight blocks (blocks of 16 here: 13 of them).  From the ninth block on the joints
    phase writes its rows as many whole turns up as the chain phase found previous_sol to be when it finished the block that used the
    workspace slot before — a hint that saves the chain phase its atomic adds on trajectories that have wound up (long runs: -5 ... -9 %)
    — and a quiet step's value is then raw + turns x 2 pi in ONE rounding instead of two: flags, states, the carried theta and the
    latch are still the same bits, joints and previous_sol the same to 1e-12 (observed: 2 ulp)."""
    A = _abi_mod()
    traj = _eventful_trajectories(torch_mod, n_traj, n_steps, 7000 + n_traj, arm)
    c = make_control()
    hs = c._solver
    forms = [("steps", A.CONT_RUN_STEPS, 0, 0), ("phased", A.CONT_RUN_PHASED, 0, 0), ("phased, events", A.CONT_RUN_PHASED, A.PHASED_EDGES_BY_EVENT, 0),
             ("phased, no theta-first", A.CONT_RUN_PHASED, A.PHASED_NO_THETA_FIRST, 0), ("phased, events, no theta-first, blocks of 48", A.CONT_RUN_PHASED, 3, 48),
             ("phased, blocks of 48", A.CONT_RUN_PHASED, 0, 48), ("phased, blocks of 16", A.CONT_RUN_PHASED, 0, 16)]
    got = {}
    for name, run_mode, variant, blk in forms:
        hs.set_option(A.OPT_CONT_RUN_MODE, run_mode)
        hs.set_option(A.OPT_CONT_PHASED_VARIANT, variant)
        hs.set_option(A.OPT_CONT_BLOCK_STEPS, blk)
        st = c.new_continuous_state(arm, n_traj)
        res = c.run_continuous_trajectories(arm, traj, st, first_step_timed_out=True, current_pose=traj[0], constrained_mode=mode,
                                            d_theta_max=dmax)
        hs.synchronize()
        got[name] = {k: v.clone() for k, v in res.items()}
        got[name]["cont_state"] = st.clone()
    hs.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_AUTO)
    hs.set_option(A.OPT_CONT_PHASED_VARIANT, 0)
    hs.set_option(A.OPT_CONT_BLOCK_STEPS, 0)
    ref = got["phased"]
    _same_run(torch_mod, {k: (v[:11] if k == "cont_state" else v) for k, v in got["steps"].items()},
              {k: (v[:11] if k == "cont_state" else v) for k, v in ref.items()}, "steps vs phased")
    for name, _, _, blk in forms[2:]:
        if blk and (n_steps + blk - 1) // blk > 8:
            # (rows 11 ... 18 of the state — the cause bits and the joints the continuity check rejected — exactly / to the same bar)
            _same_run(torch_mod, {k: (v[:11] if k == "cont_state" else v) for k, v in ref.items()},
                      {k: (v[:11] if k == "cont_state" else v) for k, v in got[name].items()}, name, joint_tol=1e-12)
            a, b = ref["cont_state"], got[name]["cont_state"]
            assert torch_mod.equal(a[11], b[11]) and float((a[12:19] - b[12:19]).abs().nan_to_num().max()) <= 1e-12, name
            continue
        for k, v in ref.items():
            a, b = v, got[name][k]
            same = torch_mod.equal(a.view(torch_mod.uint8), b.view(torch_mod.uint8))
            assert same, (name, k, float((a.double() - b.double()).abs().nan_to_num().max()))
    if n_traj >= 600:
        assert float((ref["cont_state"][9] != 0).float().mean()) > 0.02, "no trajectory latched: the rare paths were not exercised"


def test_continuous_run_outgrows_its_workspace_without_stopping_the_device(torch_mod):
    """One context, runs of growing size issued back to back WITHOUT a synchronisation in between: the workspace (and the words that tie
    the pipeline's streams) are outgrown while earlier runs are still in flight.  The call never waits for the device (round 4 did, a
    hipDeviceSynchronize inside an asynchronous API): what is outgrown is parked and freed by rsik_sync.  Every run against the step kernel."""
    from bench import make_config5_trajectories

    A = _abi_mod()
    c = make_control()
    hs = c._solver
    sizes = [(8, 40), (300, 100), (64, 17), (2000, 230), (9, 300)]
    trajs = [make_config5_trajectories(n, t, seed=77 + n) for n, t in sizes]
    hs.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_PHASED)
    runs = []
    for (n, t), traj in zip(sizes, trajs):  # (no synchronisation between these)
        st = c.new_continuous_state("r_arm", n)
        runs.append((c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0]), st))
    hs.synchronize()   # rsik_sync: also frees what was outgrown
    hs.synchronize()
    hs.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_STEPS)
    for (n, t), traj, (res, st) in zip(sizes, trajs, runs):
        st2 = c.new_continuous_state("r_arm", n)
        ref = c.run_continuous_trajectories("r_arm", traj, st2, first_step_timed_out=True, current_pose=traj[0])
        hs.synchronize()
        a = {k: v for k, v in ref.items()}
        a["cont_state"] = st2[:11]
        b = {k: v for k, v in res.items()}
        b["cont_state"] = st[:11]
        _same_run(torch_mod, a, b, ("growing", n, t))
    hs.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_AUTO)


@pytest.mark.parametrize("timed_out", [False, True])
def test_continuous_run_that_begins_latched(torch_mod, timed_out):
    """ControlIK answers previous_sol, not reachable, with the emergency state for every goal once its emergency stop is latched
    (control_ik.py:205-210), until "unfreeze".  A run whose trajectories — some of them, or all — are latched when it begins: the
    pipeline fills their steps in without walking them (the chain phase's fill_rest writes the previous_sol rows of a block on entry, stores only);
    every output and the carried state against the step kernel, launch per step; the latched trajectories' rows are previous_sol
    bit for bit and their state rows are untouched, whether or not the caller says its first step timed out."""
    A = _abi_mod()
    n_traj, n_steps = 300, 120
    c = make_control()
    hs = c._solver
    st = c.new_continuous_state("r_arm", n_traj)
    first = _eventful_trajectories(torch_mod, n_traj, n_steps, 4321, "r_arm")
    c.run_continuous_trajectories("r_arm", first, st, first_step_timed_out=True, current_pose=first[0])
    hs.synchronize()
    latched = st[9] != 0
    assert 0.02 < float(latched.double().mean()) < 0.9, "the eventful trajectories should latch some trajectories, not all"
    goals = _eventful_trajectories(torch_mod, n_traj, n_steps, 999, "r_arm")
    everyone = st.clone()
    everyone[9] = 1.0
    for start, who in ((st, latched), (everyone, torch_mod.ones_like(latched))):
        ref = None
        for run_mode, blk in ((A.CONT_RUN_STEPS, 0), (A.CONT_RUN_PHASED, 0), (A.CONT_RUN_PHASED, 48)):
            hs.set_option(A.OPT_CONT_RUN_MODE, run_mode)
            hs.set_option(A.OPT_CONT_BLOCK_STEPS, blk)
            s2 = start.clone()
            res = c.run_continuous_trajectories("r_arm", goals, s2, first_step_timed_out=timed_out, current_pose=goals[0])
            hs.synchronize()
            got = {k: v.clone() for k, v in res.items()}
            got["cont_state"] = s2[:11].clone()
            # the latched trajectories: previous_sol at every step, bit for bit; emergency state, not reachable; nothing of their state moved
            prev_sol = start[1:8, who].T.contiguous()                                      # [latched, 7]
            rows = got["joints"][:, who, :]
            assert torch_mod.equal(rows.view(torch_mod.uint8), prev_sol[None].expand_as(rows).contiguous().view(torch_mod.uint8)), (run_mode, blk)
            assert bool((got["state"][:, who] == A.STATE_EMERGENCY).all()) and bool((got["reachable"][:, who] == 0).all())
            assert torch_mod.equal(s2[:, who].view(torch_mod.uint8), start[:, who].contiguous().view(torch_mod.uint8)), (run_mode, blk)
            if ref is None:
                ref = got
            else:
                _same_run(torch_mod, ref, got, ("begins latched", run_mode, blk, timed_out), joint_tol=1e-9)
    hs.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_AUTO)
    hs.set_option(A.OPT_CONT_BLOCK_STEPS, 0)


@pytest.mark.parametrize("mode,d_theta_max,preferred", [("unconstrained", 0.01, -4 * np.pi / 6), ("low_elbow", 0.05, 0.5)])
def test_config5_full_size_against_checker(torch_mod, orc, mode, d_theta_max, preferred):
    """BASELINE config 5 at its stated size: 4096 trajectories x 1000 control steps in one rsik_control_continuous_run
    call (bench.py's generator and call), a 96-trajectory subsample re-walked step by step by the CPU checker's state
    machine: flags and state codes exact, joints <= 1e-7 at every one of the 96 000 trajectory-steps, and the carried
    state at the end; the other trajectories through size-independent properties (finite joints everywhere, every step
    within the continuity thresholds of control_ik.py:398 from its predecessor, no emergency stop)."""
    from bench import URDF as BENCH_URDF
    from bench import make_config5_trajectories

    n_traj, n_steps = 4096, 1000
    traj = make_config5_trajectories(n_traj, n_steps, seed=20250204)
    c = make_control()
    assert BENCH_URDF == URDF
    st = c.new_continuous_state("r_arm", n_traj)
    res = c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0],
                                        constrained_mode=mode, d_theta_max=d_theta_max, preferred_theta=preferred)
    torch_mod.cuda.synchronize()
    J = res["joints"].cpu().numpy()
    F = res["reachable"].cpu().numpy()
    S = res["state"].cpu().numpy()
    assert np.isfinite(J).all()
    if mode == "unconstrained":
        assert st[9].sum().item() == 0
    live = (S != 8)
    step = np.abs(np.diff(J, axis=0))
    assert step[..., :4].max() <= 0.5 and step[..., 4:].max() <= 1.0
    frac = F[live].mean()
    assert 0.2 < frac < 0.5  # 66 % of the steps take the unreachable fallback on this generator
    sub = np.arange(0, n_traj, n_traj // 96)[:96]
    M12 = traj[:, :, torch_mod.as_tensor(sub).cuda()].cpu().numpy()          # [n_steps, 12, 96]
    a = orc.Arm("r_arm", -1.01)
    worst = 0.0
    for col, k in enumerate(sub):
        cs = orc.ContinuousState(c.previous_theta["r_arm"], c.previous_sol["r_arm"])
        Ms = np.tile(np.eye(4), (n_steps, 1, 1))
        Ms[:, :3, :3] = M12[:, :9, col].reshape(n_steps, 3, 3)
        Ms[:, :3, 3] = M12[:, 9:, col]
        for i in range(n_steps):
            j, ok, code = orc.control_continuous_step(a, cs, Ms[i], timed_out=(i == 0), preferred_theta_arg=preferred,
                                                      preferred_theta_self=c.preferred_theta["r_arm"],
                                                      constrained_mode=_abi_mod().MODES[mode],
                                                      current_joints=cs.previous_sol, current_pose=Ms[0], d_theta_max=d_theta_max)
            assert ok == bool(F[i, k]) and code == S[i, k], (k, i)
            worst = max(worst, float(np.max(np.abs(j - J[i, k]))))
        assert abs(cs.previous_theta - float(st[0, k])) < 1e-9
        assert cs.emergency_stop == bool(st[9, k] != 0)
    assert worst < 1e-7, worst


# ------------------------------------------------------------------------------------------ emergency diagnostics (G11)
def _same_report(a, b, tol=1e-7):
    """Two emergency_state texts: the words must be identical, the numbers NumPy printed (8 significant digits) equal to
    within `tol` (the drop-in's joints differ from the reference's by ~1e-12, which can flip a last printed digit)."""
    import re

    num = re.compile(r"[-+]?\d+\.\d*(?:e[-+]?\d+)?|[-+]?\d+e[-+]?\d+")
    ta, tb = num.sub("#", a), num.sub("#", b)
    if re.sub(r"\s+", " ", ta) != re.sub(r"\s+", " ", tb):
        return False
    na, nb = [float(x) for x in num.findall(a)], [float(x) for x in num.findall(b)]
    return len(na) == len(nb) and all(abs(x - y) <= tol + 1e-7 * abs(y) for x, y in zip(na, nb))


def test_scale_continuous_digests_of_the_reference(golden_dir, torch_mod):
    """G16 — config 5's generator through the reference ITSELF: 512 trajectories x 1000 control steps walked by ControlIK objects in
    the build container (oracle/gen_golden.py gen_scale_continuous), digests of the `reachable` and `state` arrays, sixteen
    trajectories' joints and the carried theta.  The trajectory pipeline over the regenerated matrices (tests/scale_inputs.py): flags
    and state codes bit-exact at half a million control steps, joints <= 1e-7, theta <= 1e-9 — no checker in between."""
    from tests import scale_inputs as SC
    from tests.test_oracle_golden import _check_scale_continuous

    g = load(golden_dir, "g16_scale_continuous.npz")
    M = SC.config5_trajectories()
    assert SC.sha256(M) == str(g["input_sha256"]), "the seeded inputs did not regenerate"
    n_steps, n_traj = M.shape[:2]
    c = make_control()
    for run_mode in (_abi_mod().CONT_RUN_PHASED, _abi_mod().CONT_RUN_STEPS):
        c._solver.set_option(_abi_mod().OPT_CONT_RUN_MODE, run_mode)
        st = c.new_continuous_state("r_arm", n_traj)
        Mt = torch_mod.as_tensor(M).cuda()
        start = np.tile(np.asarray(c.previous_pose["r_arm"], dtype=np.float64), (n_traj, 1, 1))  # (a fresh ControlIK's previous_pose)
        res = to_np(c.run_continuous_trajectories("r_arm", Mt, st, first_step_timed_out=True, current_pose=start))
        torch_mod.cuda.synchronize()
        _check_scale_continuous(g, res, st[0].cpu().numpy(), st[9].cpu().numpy())
    assert 0 < int(g["emergency_stop"].sum()) < 8  # (one trajectory trips the reference's continuity check at the gimbal lock and stays latched)
    c._solver.set_option(_abi_mod().OPT_CONT_RUN_MODE, _abi_mod().CONT_RUN_AUTO)


def test_scale_variants_digests_of_the_reference(golden_dir, torch_mod):
    """G17 — the other arm, the other modes, at scale, against the reference itself: ControlIK discrete on the l_arm mirror images of
    config 3's 256 Ki goal matrices with is_dvt=True (the kernels' singularity-plane variant), "low_elbow", 20 grid points; ControlIK
    continuous on the l_arm mirror images of 256 of G16's trajectories x 1000 steps, "low_elbow", d_theta_max = 0.05 (pipeline and
    step kernel).  Flags and state codes by digest, joints on the subsamples."""
    from tests import scale_inputs as SC
    from tests.test_oracle_golden import _check_scale_continuous, _check_scale_set

    A = _abi_mod()
    g = load(golden_dir, "g17_scale_variants.npz")
    _, _, _, Mr = SC.config3_from_kept(load(golden_dir, "g14_scale.npz")["c3_kept_bits"])
    Ml = SC.mirror_matrices(Mr)
    assert SC.sha256(Ml) == str(g["d_input_sha256"]), "the seeded inputs did not regenerate"
    c = make_control(is_dvt=True)
    c.nb_search_points = 20
    res = to_np(c.symbolic_inverse_kinematics_batch("l_arm", Ml, constrained_mode="low_elbow"))
    _check_scale_set(g, "d_", res, len(Ml))
    Mt = SC.mirror_matrices(SC.config5_trajectories()[:, :256])
    assert SC.sha256(Mt) == str(g["input_sha256"]), "the seeded inputs did not regenerate"
    n_traj = Mt.shape[1]
    c = make_control()
    for run_mode in (A.CONT_RUN_PHASED, A.CONT_RUN_STEPS):
        c._solver.set_option(A.OPT_CONT_RUN_MODE, run_mode)
        st = c.new_continuous_state("l_arm", n_traj)
        start = np.tile(np.asarray(c.previous_pose["l_arm"], dtype=np.float64), (n_traj, 1, 1))
        res = to_np(c.run_continuous_trajectories("l_arm", torch_mod.as_tensor(Mt).cuda(), st, first_step_timed_out=True, current_pose=start,
                                                  constrained_mode="low_elbow", d_theta_max=0.05))
        torch_mod.cuda.synchronize()
        _check_scale_continuous(g, res, st[0].cpu().numpy(), st[9].cpu().numpy())
    c._solver.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_AUTO)


def test_emergency_reports_discrete(golden_dir, torch_mod):
    """G11: previous_sol next to the +-6 pi multiturn limit (utils.multiturn_safety_check, utils.py:535-568), every
    joint / sign / combination: the call that trips returns the clamped joints with its normal state, latches the stop
    and appends the reference's messages to emergency_state; the FOLLOWING call returns (previous_sol, False,
    emergency_state) (control_ik.py:205-210)."""
    import contextlib
    import io

    from reachy2_symbolic_ik_amd import ControlIK

    g = load(golden_dir, "g11_emergency.npz")
    for arm in ("r_arm", "l_arm"):
        pre = f"{arm}_discrete_"
        for k in range(len(g[pre + "M"])):
            with contextlib.redirect_stdout(io.StringIO()):
                c = ControlIK(current_joints=g[pre + "current_joints"][k].tolist(), urdf_path=URDF)
                j1, ok1, st1 = c.symbolic_inverse_kinematics(arm, g[pre + "M"][k], "discrete")
            assert ok1 == bool(g[pre + "ok1"][k]) and st1 == str(g[pre + "state1"][k])
            assert np.max(np.abs(np.asarray(j1) - g[pre + "joints1"][k])) < TOL
            assert c.emergency_stop and c.emergency_state == str(g[pre + "emergency_state"][k]), (arm, k, c.emergency_state)
            with contextlib.redirect_stdout(io.StringIO()):
                j2, ok2, st2 = c.symbolic_inverse_kinematics(arm, g[pre + "M"][k], "discrete")
            assert not ok2 and st2 == str(g[pre + "state2"][k])
            assert np.max(np.abs(np.asarray(j2) - g[pre + "joints2"][k])) < TOL
        # the batch entry point reports the same causes as bits
        c = make_control()
        c.previous_sol[arm] = np.array(g[pre + "current_joints"][0][("r_arm", "l_arm").index(arm)])
        res = to_np(c.symbolic_inverse_kinematics_batch(arm, g[pre + "M"][:1]))
        assert res["emergency"][0] == g[pre + "cause"][0]


def test_emergency_reports_continuous(golden_dir, torch_mod, monkeypatch):
    """G11, continuous mode with the reference's clock patched like the generator patched it: a goal jump trips
    utils.continuity_check (utils.py:571-589: the message quotes previous_joints and the rejected joints), a start
    next to +-6 pi trips multiturn_safety_check; the stop latches, the following calls return emergency_state, and
    "unfreeze" clears it (control_ik.py:196-210)."""
    import contextlib
    import io

    import reachy2_symbolic_ik_amd.control_ik as cik
    from reachy2_symbolic_ik_amd import ControlIK

    g = load(golden_dir, "g11_emergency.npz")

    class Clock:
        t = 1000.0

        @staticmethod
        def time():
            return Clock.t

    monkeypatch.setattr(cik, "time", Clock)
    for arm in ("r_arm", "l_arm"):
        pre = f"{arm}_continuous_"
        Ms, J, OK, ST, ES, CT = (g[pre + k] for k in ("M", "joints", "ok", "state", "emergency_state", "control_type"))
        for k in range(Ms.shape[0]):
            Clock.t = 1000.0
            with contextlib.redirect_stdout(io.StringIO()):
                c = ControlIK(urdf_path=URDF)
            for i in range(Ms.shape[1]):
                Clock.t += 1.0 / 120.0
                start = g[pre + "start_joints"][k]
                kw = dict(current_joints=list(start)) if (i == 0 and not np.isnan(start[0])) else {}
                with contextlib.redirect_stdout(io.StringIO()):
                    j, ok, st = c.symbolic_inverse_kinematics(arm, Ms[k, i], str(CT[k, i]), d_theta_max=0.01, **kw)
                assert ok == bool(OK[k, i]), (arm, k, i)
                assert _same_report(st, str(ST[k, i])), (arm, k, i, st, str(ST[k, i]))
                assert _same_report(c.emergency_state, str(ES[k, i])), (arm, k, i, c.emergency_state)
                assert c.emergency_stop == bool(g[pre + "estop"][k, i])
                assert np.max(np.abs(np.asarray(j) - J[k, i])) < 1e-7, (arm, k, i)
                assert np.max(np.abs(np.asarray(c.previous_sol[arm]) - g[pre + "previous_sol"][k, i])) < 1e-7
        # the same trajectories as a batch (up to the "unfreeze" call): both ways rsik_control_continuous_run issues
        # the work, state codes with the latched steps reported as RSIK_STATE_EMERGENCY, reports from the state rows
        A = _abi_mod()
        upto = Ms.shape[1] - 3
        codes = {s: k for k, s in enumerate(__import__("reachy2_symbolic_ik_amd").STATE_STRINGS)}
        for run_mode in (A.CONT_RUN_PHASED, A.CONT_RUN_STEPS):
            c = make_control()
            c._solver.set_option(A.OPT_CONT_RUN_MODE, run_mode)
            sel = np.isnan(g[pre + "start_joints"][:, 0])  # default start (the jump scenarios)
            st = c.new_continuous_state(arm, int(sel.sum()))
            start_pose = np.tile(np.asarray(c.previous_pose[arm], dtype=np.float64), (int(sel.sum()), 1, 1))
            res = to_np(c.run_continuous_trajectories(arm, np.swapaxes(Ms[sel][:, :upto], 0, 1), st, first_step_timed_out=True,
                                                      current_pose=start_pose))
            rep = c.emergency_report(st)
            for col, k in enumerate(np.nonzero(sel)[0]):
                for i in range(upto):
                    assert res["reachable"][i, col] == OK[k, i]
                    want = codes.get(str(ST[k, i]), 8)
                    assert res["state"][i, col] == want, (arm, k, i)
                    assert np.max(np.abs(res["joints"][i, col] - J[k, i])) < 1e-7
                assert bool(g[pre + "estop"][k, upto - 1]) == (col in rep)
                if col in rep:
                    assert _same_report(rep[col], str(ES[k, upto - 1])), (arm, k, rep[col])


# ------------------------------------------------------------------------------------------ oracle-free properties
def test_fk_of_ik_is_identity_full_size(torch_mod):
    """Size-independent, checker-free property at the BASELINE size: forward kinematics of the solved joints gives the
    goal pose back (1 M poses, any theta inside the interval), wherever the solver does not move the goal (no backward
    wrist shift, no min-distance reduce; singularity_offset = -1.01 so no elbow projection)."""
    from bench import make_config2_poses
    from tests.fk_numpy import forward_kinematics

    pos, eul = make_config2_poses(1 << 20, seed=4242)
    solver, r, l = make_symbolic(-1.01)
    rng = np.random.default_rng(9)
    frac = torch_mod.as_tensor(rng.uniform(0.02, 0.98, size=len(pos))).cuda()
    res = to_np(r.solve_batch(soa(pos, eul, torch_mod), theta=("fraction", frac)))
    assert res["reachable"].all()
    from reachy2_symbolic_ik_amd.constants import euler_xyz_extrinsic
    ca, sa, cb, sb, cc, sc = (f(eul[:, k]) for k in (0, 1, 2) for f in (np.cos, np.sin))
    Rg = np.empty((len(pos), 3, 3))
    Rg[:, 0, 0] = cc * cb; Rg[:, 0, 1] = cc * sb * sa - sc * ca; Rg[:, 0, 2] = cc * sb * ca + sc * sa
    Rg[:, 1, 0] = sc * cb; Rg[:, 1, 1] = sc * sb * sa + cc * ca; Rg[:, 1, 2] = sc * sb * ca - cc * sa
    Rg[:, 2, 0] = -sb; Rg[:, 2, 1] = cb * sa; Rg[:, 2, 2] = cb * ca
    assert np.allclose(Rg[0], euler_xyz_extrinsic(eul[0]))
    s = np.array([0.0, -0.2, 0.0])
    w = pos + 0.1 * Rg[:, :, 2]
    dsw = np.linalg.norm(w - s, axis=1)
    untouched = (w[:, 0] >= 0.02 + 1e-9) & (dsw >= r.shoulder_wrist_min_distance + 1e-9)
    assert untouched.mean() > 0.8
    p_fk, R_fk = forward_kinematics(res["joints"][untouched], s, [-15, 0, 10], 0.28, 0.28, 0.10)
    assert np.max(np.abs(p_fk - pos[untouched])) < 1e-9
    assert np.max(np.abs(R_fk - Rg[untouched])) < 1e-8


# ------------------------------------------------------------------------------------------ on-device FK
@pytest.mark.parametrize("arm", ["r_arm", "l_arm"])
def test_device_forward_kinematics_matches_numpy_chain(torch_mod, arm):
    """rsik_forward_kinematics against the NumPy restatement of the same chain (tests/fk_numpy.py) on random joints."""
    from tests.fk_numpy import forward_kinematics

    solver, r, l = make_symbolic(0.03)
    ik = r if arm == "r_arm" else l
    rng = np.random.default_rng(77)
    j = rng.uniform(-np.pi, np.pi, size=(5000, 7))
    pos, rot = ik.forward_kinematics_batch(j)
    s = np.array([0.0, -0.2, 0.0]) if arm == "r_arm" else np.array([0.0, 0.2, 0.0])
    off = [-15, 0, 10] if arm == "r_arm" else [15, 0, -10]
    p_ref, R_ref = forward_kinematics(j, s, off, 0.28, 0.28, 0.10)
    assert np.max(np.abs(pos.cpu().numpy() - p_ref)) < 1e-14
    assert np.max(np.abs(rot.cpu().numpy() - R_ref)) < 1e-14


def test_device_fk_residual_full_size(torch_mod):
    """FK(IK(pose)) == pose on the device for 1 M reachable poses (pose and matrix goals), wherever the solver leaves
    the goal where it was; projected / shifted goals show up as a residual equal to the shift, never as NaN."""
    from bench import make_config2_poses

    pos, eul = make_config2_poses(1 << 20, seed=777)
    solver, r, l = make_symbolic(-1.01)
    s6 = soa(pos, eul, torch_mod)
    res = r.solve_batch(s6)
    err = r.fk_residual_batch(s6, res["joints"]).cpu().numpy()
    assert np.isfinite(err).all()
    assert np.quantile(err[:, 0], 0.8) < 1e-12 and np.quantile(err[:, 1], 0.8) < 1e-12
    # a moved goal (backward wrist shift <= 0.1 m, min-distance reduce <= 0.25 m) keeps its orientation, up to the
    # elbow-pitch clamp that the 1e-8 projection margin of the min-distance reduce triggers (symbolic_ik.py:166-171, 853)
    assert err[:, 0].max() < 0.4 and err[:, 1].max() < 1e-5
    same = err[:, 0] < 1e-9
    assert same.mean() > 0.8 and err[same, 1].max() < 1e-8
    # the same residual through the matrix form of the goal
    from reachy2_symbolic_ik_amd.constants import euler_xyz_extrinsic

    n = 4096
    M = np.zeros((12, n))
    for k in range(n):
        M[:9, k] = euler_xyz_extrinsic(eul[k]).reshape(9)
    M[9:, :] = pos[:n].T
    err12 = solver.fk_residual(torch_mod.as_tensor(M).cuda(), res["joints"][:n].contiguous(), arm_uniform=0).cpu().numpy()
    assert np.max(np.abs(err12 - err[:n])) < 1e-12
    # unreachable rows: NaN in, NaN out
    bad = torch_mod.full((8, 7), float("nan"), dtype=torch_mod.float64, device="cuda")
    e = solver.fk_residual(s6[:, :8].contiguous(), bad, arm_uniform=0)
    assert bool(torch_mod.isnan(e).all())


# ------------------------------------------------------------------ goal matrix <-> Euler pose
MATRIX_KINDS = ("proper", "gimbal", "skewed", "near_identity")


def _euler_to_matrix(e):
    ca, sa, cb, sb, cc, sc = np.cos(e[:, 0]), np.sin(e[:, 0]), np.cos(e[:, 1]), np.sin(e[:, 1]), np.cos(e[:, 2]), np.sin(e[:, 2])
    return np.stack([cc * cb, cc * sb * sa - sc * ca, cc * sb * ca + sc * sa, sc * cb, sc * sb * sa + cc * ca, sc * sb * ca - cc * sa,
                     -sb, cb * sa, cb * ca], axis=1).reshape(-1, 3, 3)


@pytest.mark.parametrize("kind", MATRIX_KINDS)
def test_matrix_to_pose_golden(golden_dir, torch_mod, kind):
    """rsik_matrix_to_pose = batched utils.get_euler_from_homogeneous_matrix (utils.py:84-90) against the angles the
    reference returned (G8): proper rotations, gimbal lock (third angle := 0 within 1e-7 of the lock), matrices that
    are not orthonormal (nearest rotation first), near-identity matrices."""
    g = load(golden_dir, "g8_matrix_edges.npz")
    c = make_control()
    for arm in ("r_arm", "l_arm"):
        pre = f"{arm}_{kind}_"
        M = g[pre + "M"]
        pose = c.matrices_to_poses(M, identity_shortcut=False).cpu().numpy()
        np.testing.assert_array_equal(pose[:3].T, M[:, :3, 3])
        eul, ref = pose[3:].T, g[pre + "euler"]
        if kind == "gimbal":
            assert np.max(np.abs(_euler_to_matrix(eul) - _euler_to_matrix(ref))) < 1e-9
            locked = np.abs(np.abs(ref[:, 1]) - np.pi / 2) < 5e-8
            assert locked.sum() > 50 and np.all(eul[locked, 2] == 0.0)
            free = np.abs(np.abs(ref[:, 1]) - np.pi / 2) > 1.2e-7
            assert free.sum() > 50 and np.all(eul[free, 2] != 0.0) and np.all(ref[free, 2] != 0.0)
        else:
            assert np.max(np.abs(eul - ref)) < 1e-12
        # ControlIK's shortcut (control_ik.py:212-214): allclose(R, I) => angles exactly 0
        if kind == "near_identity":
            snap = c.matrices_to_poses(M, identity_shortcut=True).cpu().numpy()[3:].T
            close = np.array([np.allclose(m[:3, :3], np.eye(3)) for m in M])
            assert close.sum() > 20 and (~close).sum() > 20
            assert np.all(snap[close] == 0.0) and np.max(np.abs(snap[~close] - ref[~close])) < 1e-12


@pytest.mark.parametrize("kind", MATRIX_KINDS)
def test_control_discrete_matrix_edges(golden_dir, torch_mod, kind):
    """ControlIK discrete on the G8 matrices.  The kernels consume the rotation directly and make the reference's
    matrix -> Euler -> matrix round trip only where it changes the result ("auto", the default): that must reproduce
    the reference everywhere, like "always".  "never" shows what the round trip is worth: ~4e-6 rad at gimbal lock,
    ~1e-3 on the deliberately skewed matrices."""
    g = load(golden_dir, "g8_matrix_edges.npz")
    for mode in ("auto", "always", "never"):
        c = make_control()
        c.euler_roundtrip = mode
        for arm in ("r_arm", "l_arm"):
            pre = f"{arm}_{kind}_"
            res = to_np(c.symbolic_inverse_kinematics_batch(arm, g[pre + "M"]))
            if mode == "never" and kind in ("skewed", "gimbal"):
                same = (res["reachable"] == g[pre + "reachable"]) & (g[pre + "reachable"] == 1)
                assert same.mean() > 0.3
                err = np.max(np.abs(res["joints"][same] - g[pre + "joints"][same]))
                assert err > (1e-5 if kind == "skewed" else 1e-7), f"{pre}: the round trip made no difference ({err})"
                continue
            np.testing.assert_array_equal(res["reachable"], g[pre + "reachable"])
            np.testing.assert_array_equal(res["state"], g[pre + "state"])
            err = np.max(np.abs(res["joints"] - g[pre + "joints"]))
            assert err < TOL, f"{pre} mode={mode}: {err}"


def test_host_resident_pipeline_matches_device_path(torch_mod):
    """SymbolicIK.solve_batch_host (host-resident batch, chunked uploads / solves / downloads on three streams) returns
    exactly what one device-resident solve_batch returns, including a ragged last chunk and all outcome states."""
    solver, r, l = make_symbolic(0.03)
    rng = np.random.default_rng(5)
    n = 3 * 4096 + 777
    pos = np.array([0.0, -0.2, 0.0]) + rng.uniform(-0.7, 0.7, size=(n, 3))
    eul = rng.uniform(-np.pi, np.pi, size=(n, 3))
    host = torch_mod.as_tensor(np.ascontiguousarray(np.concatenate([pos.T, eul.T], axis=0)))
    ref = to_np(r.solve_batch(host.cuda(), want_elbow=False))
    for src in (host, host.pin_memory()):
        res = {k: v.numpy() for k, v in r.solve_batch_host(src, chunk=4096).items()}
        for k in ("reachable", "state"):
            np.testing.assert_array_equal(res[k], ref[k])
        for k in ("joints", "interval"):
            np.testing.assert_array_equal(res[k], ref[k])  # same kernel, same inputs: bit-identical (NaN rows included)
    assert r.solve_batch_host(host[:, :0])["joints"].shape == (0, 7)


def test_custom_geometry_golden(golden_dir, torch_mod):
    """G9: non-default arm (tip with x / y components, u != f, other shoulder offsets, elbow / wrist / backward limits
    and singularity plane) recorded from the reference with the same constructor arguments: the kernels' general
    formulas, not only their default-geometry special case.  Uniform launches per arm and one mixed launch."""
    import contextlib
    import io

    from test_oracle_golden import CUSTOM_GEOMETRY

    from reachy2_symbolic_ik_amd import DualArmIK, HipSolver, SymbolicIK

    g = load(golden_dir, "g9_custom_geometry.npz")
    solver = HipSolver(0)
    with contextlib.redirect_stdout(io.StringIO()):
        iks = {arm: SymbolicIK(arm, solver=solver, **CUSTOM_GEOMETRY) for arm in ("r_arm", "l_arm")}
        dual = DualArmIK(solver=solver, **CUSTOM_GEOMETRY)
    for arm, ik in iks.items():
        for f in ("gripper_size", "max_arm_length", "shoulder_wrist_min_distance", "elbow_singularity_position"):
            np.testing.assert_allclose(getattr(ik, f), g[f"{arm}_const_{f}"], rtol=0, atol=1e-15)
        check_symbolic(to_np(ik.solve_batch(soa(g[f"{arm}_sweep_pos"], g[f"{arm}_sweep_eul"], torch_mod))), g, f"{arm}_sweep_")
        p = soa(g[f"{arm}_reach_pos"], g[f"{arm}_reach_eul"], torch_mod)
        check_symbolic(to_np(ik.solve_batch(p)), g, f"{arm}_reach_i0_")
        tu = torch_mod.as_tensor(g[f"{arm}_reach_theta_u"]).cuda()
        check_symbolic(to_np(ik.solve_batch(p, theta=("fraction", tu))), g, f"{arm}_reach_in_")
    # both arms in one mixed launch
    pos = np.concatenate([g["r_arm_reach_pos"], g["l_arm_reach_pos"]])
    eul = np.concatenate([g["r_arm_reach_eul"], g["l_arm_reach_eul"]])
    n = len(g["r_arm_reach_pos"])
    arm_id = np.concatenate([np.zeros(n, np.uint8), np.ones(n, np.uint8)])
    res = to_np(dual.solve_batch(torch_mod.as_tensor(arm_id).cuda(), soa(pos, eul, torch_mod)))
    merged = {f"m_{k}": np.concatenate([g[f"r_arm_reach_i0_{k}"], g[f"l_arm_reach_i0_{k}"]]) for k in
              ("reachable", "state", "interval", "joints", "elbow")}
    check_symbolic(res, merged, "m_")


def test_tip_z_specialisation_matches_general_path(torch_mod):
    """The default arm's tip offset has no x / y component, so rsik_solve launches the specialised stages
    (goal_from_euler_tipz; no renormalisation of the wrist-yaw direction, which is a unit vector by construction then).
    RSIK_OPT_NO_TIPZ forces the general path: flags, intervals, elbows and the first six joints must be the same bits,
    the wrist yaw the same to rounding, for uniform and mixed launches, on all outcomes."""
    from reachy2_symbolic_ik_amd import DualArmIK

    solver, r, l = make_symbolic(0.03)
    rng = np.random.default_rng(12)
    n = 50000
    pos = np.array([0.0, 0.0, 0.0]) + rng.uniform(-0.7, 0.7, size=(n, 3))
    eul = rng.uniform(-np.pi, np.pi, size=(n, 3))
    arm_id = torch_mod.as_tensor((rng.uniform(size=n) < 0.5).astype(np.uint8)).cuda()
    p = soa(pos, eul, torch_mod)
    import contextlib
    import io

    with contextlib.redirect_stdout(io.StringIO()):
        dual = DualArmIK(solver=solver)
    fast = [to_np(r.solve_batch(p)), to_np(dual.solve_batch(arm_id, p))]
    solver.set_option(_abi_mod().OPT_NO_TIPZ, 1)
    slow = [to_np(r.solve_batch(p)), to_np(dual.solve_batch(arm_id, p))]
    for a, b in zip(fast, slow):
        assert a["reachable"].sum() > 500
        for k in ("reachable", "state", "interval", "elbow"):
            np.testing.assert_array_equal(a[k], b[k])
        np.testing.assert_array_equal(a["joints"][:, :6], b["joints"][:, :6])
        m = a["reachable"].astype(bool)
        assert np.max(np.abs(a["joints"][m, 6] - b["joints"][m, 6])) < 2e-15
    # mixed launches of mirror-image arms read the constants without handedness as scalars (RSIK_OPT_NO_MIRROR: all per
    # lane from LDS): the same bits either way
    solver.set_option(_abi_mod().OPT_NO_TIPZ, 0)
    solver.set_option(_abi_mod().OPT_NO_MIRROR, 1)
    c = to_np(dual.solve_batch(arm_id, p))
    for k in ("reachable", "state", "interval", "joints", "elbow"):
        np.testing.assert_array_equal(c[k], fast[1][k])


def test_solve_is_hipgraph_capturable(torch_mod, orc):
    """the launches are plain kernel launches on the context's stream, so a caller can record them in a
    hipGraph and replay it on new data in the same buffers (what bench.py does for its K timed steps)."""
    solver, r, l = make_symbolic(0.03)
    rng = np.random.default_rng(21)
    n = 5000

    def batch():
        pos = np.array([0.0, -0.2, 0.0]) + rng.uniform(-0.7, 0.7, size=(n, 3))
        return pos, rng.uniform(-np.pi, np.pi, size=(n, 3))

    pos, eul = batch()
    buf = soa(pos, eul, torch_mod)
    out = {"joints": torch_mod.empty((n, 7), dtype=torch_mod.float64, device="cuda"),
           "interval": torch_mod.empty((n, 2), dtype=torch_mod.float64, device="cuda"),
           "reachable": torch_mod.empty((n,), dtype=torch_mod.uint8, device="cuda"),
           "state": torch_mod.empty((n,), dtype=torch_mod.uint8, device="cuda")}
    plan = r.solve_batch(buf, want_elbow=False, out=out, plan_only=True)
    plan["launch"]()
    torch_mod.cuda.synchronize()
    g = torch_mod.cuda.CUDAGraph()
    with torch_mod.cuda.graph(g, capture_error_mode="thread_local"):
        cs = torch_mod.cuda.current_stream().cuda_stream  # a planned launch goes to its planning stream unless told otherwise
        plan["launch"](cs)
        plan["launch"](cs)
    for _ in range(2):
        pos, eul = batch()
        buf.copy_(soa(pos, eul, torch_mod))
        for k in out:
            out[k].zero_()
        g.replay()
        torch_mod.cuda.synchronize()
        ref = orc.solve_batch(orc.Arm("r_arm", 0.03), orc.Arm("l_arm", 0.03), pos, eul)
        np.testing.assert_array_equal(out["reachable"].cpu().numpy(), ref["reachable"])
        np.testing.assert_array_equal(out["state"].cpu().numpy(), ref["state"])
        m = ref["reachable"].astype(bool)
        assert m.sum() > 100 and np.max(np.abs(out["joints"].cpu().numpy()[m] - ref["joints"][m])) < TOL


def test_c_program_through_the_abi(tmp_path, torch_mod):
    """examples/solve_from_c.c: a plain C host (no HIP headers, no Python, no torch in the process) drives the library
    through include/rsik.h alone and gets what the Python drop-in class gets."""
    import shutil
    import subprocess

    from reachy2_symbolic_ik_amd import _abi

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    solver, r, l = make_symbolic(0.03)
    consts = tmp_path / "consts.bin"
    np.ascontiguousarray(r.consts, dtype=np.float64).tofile(consts)
    exe = tmp_path / "solve_from_c"
    libdir = os.path.dirname(_abi.LIB_PATH)
    subprocess.check_call([shutil.which("gcc"), "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "solve_from_c.c"), "-o", str(exe), "-L", libdir, "-lrsik_hip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm"])
    out = subprocess.run([str(exe), str(consts)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = np.array([[float(x) for x in line.split()] for line in out.stdout.strip().splitlines()])
    pos = np.array([[0.55, -0.3, -0.15], [0.3, -0.1, 0.1], [-0.3, -0.2, 0.0], [1.5, -0.2, 0.0]])
    eul = np.array([[0, -np.pi / 2, 0], np.radians([20, -50, 20]), [0, 0, 0], [0, 0, 0]])
    ref = to_np(r.solve_batch(soa(pos, eul, torch_mod)))
    np.testing.assert_array_equal(rows[:, 0], ref["reachable"])
    np.testing.assert_array_equal(rows[:, 1], ref["state"])
    assert list(rows[:, 0]) == [1, 1, 0, 0] and list(rows[:, 1]) == [0, 0, 2, 1]  # reachable x2, backward, out of reach
    np.testing.assert_array_equal(rows[:, 2:4], ref["interval"])
    np.testing.assert_array_equal(rows[:, 4:11], ref["joints"])


def test_custom_urdf_control_golden(golden_dir, torch_mod):
    """G10: ControlIK(urdf=<a URDF that is not the Reachy 2 one>) — URDF -> parameters -> constants -> kernels for a
    non-default geometry: discrete mode (two grid sizes / constrained modes, uniform and mixed launches) and continuous
    trajectories (trajectory mode, both arms in one mixed launch)."""
    import contextlib
    import io

    from reachy2_symbolic_ik_amd import ControlIK

    g = load(golden_dir, "g10_custom_urdf_control.npz")
    urdf = open(os.path.join(golden_dir, "custom_arm.urdf")).read()
    with contextlib.redirect_stdout(io.StringIO()):
        c = ControlIK(urdf=urdf)
    for arm in ("r_arm", "l_arm"):
        s = c.symbolic_ik_solver[arm]
        np.testing.assert_array_equal(s.shoulder_position, g[f"{arm}_param_shoulder_position"])
        np.testing.assert_array_equal(np.asarray(s.shoulder_orientation_offset, dtype=float), g[f"{arm}_param_shoulder_orientation_offset"])
        np.testing.assert_array_equal(s.tip_position, g[f"{arm}_param_tip_position"])
        assert s.upper_arm_size == g[f"{arm}_param_upper_arm_size"] and s.forearm_size == g[f"{arm}_param_forearm_size"]
    for key, nb, mode in (("u20", 20, "unconstrained"), ("l64", 64, "low_elbow")):
        c.nb_search_points = nb
        for arm in ("r_arm", "l_arm"):
            res = to_np(c.symbolic_inverse_kinematics_batch(arm, g[f"{arm}_M"], constrained_mode=mode))
            np.testing.assert_array_equal(res["reachable"], g[f"{arm}_{key}_reachable"])
            np.testing.assert_array_equal(res["state"], g[f"{arm}_{key}_state"])
            assert np.max(np.abs(res["joints"] - g[f"{arm}_{key}_joints"])) < TOL
        M = np.concatenate([g["r_arm_M"], g["l_arm_M"]])
        arm_id = torch_mod.as_tensor(np.concatenate([np.zeros(len(g["r_arm_M"]), np.uint8), np.ones(len(g["l_arm_M"]), np.uint8)])).cuda()
        res = to_np(c.symbolic_inverse_kinematics_batch(arm_id, M, constrained_mode=mode))
        np.testing.assert_array_equal(res["state"], np.concatenate([g[f"r_arm_{key}_state"], g[f"l_arm_{key}_state"]]))
        assert np.max(np.abs(res["joints"] - np.concatenate([g[f"r_arm_{key}_joints"], g[f"l_arm_{key}_joints"]]))) < TOL
    cat = lambda k: np.concatenate([g[f"r_arm_traj_{k}"], g[f"l_arm_traj_{k}"]])  # noqa: E731
    Ms, J, F, S = cat("M"), cat("joints"), cat("reachable"), cat("state")
    nt = len(g["r_arm_traj_M"])
    arm_id = torch_mod.as_tensor(np.concatenate([np.zeros(nt, np.uint8), np.ones(nt, np.uint8)])).cuda()
    st = c.new_continuous_state(arm_id, Ms.shape[0])
    res = to_np(c.run_continuous_trajectories(arm_id, np.swapaxes(Ms, 0, 1), st, first_step_timed_out=True,
                                              current_joints=cat("start_joints"), current_pose=cat("start_pose")))
    np.testing.assert_array_equal(res["reachable"], F.T)
    np.testing.assert_array_equal(res["state"], S.T)
    assert np.max(np.abs(res["joints"] - np.swapaxes(J, 0, 1))) < 1e-7


def test_interval_closed_form_hands_over_at_decision_boundaries(torch_mod, orc):
    """reach: the theta interval comes from a closed form ([phi - alpha, phi + alpha]) except where rounding decides the
    outcome, which is left to the reference's own arithmetic (tangency |R'^2 - D'^2| < 1e-8, near-parallel planes, ...).
    Poses are walked across the reachable / "limited by wrist" boundary (found by bisection on the checker, to the last
    bit of the pitch angle) at offsets from 1e-2 down to 1e-13 rad on both sides: flags must stay bit-exact and the
    interval within tolerance through the hand-over between the two code paths.  (Within ~1e-15 rad of the boundary —
    a few ulps of the angle — the outcome is decided by the last bit of sin / cos of the input angle itself, where the
    kernels' sincos, within 3e-16 of libm, is not bit-identical to NumPy's: 60 of 160 such poses differ, with the old
    code path as with the new one; that is outside what "bit-exact flags" can mean for a different sincos.)"""
    solver, r, l = make_symbolic(0.03)
    ar, al = orc.Arm("r_arm", 0.03), orc.Arm("l_arm", 0.03)
    rng = np.random.default_rng(77)
    base_pos, base_eul, lo_hi = [], [], []
    # pairs (reachable pose, limited-by-wrist pose) that differ in the pitch angle only
    while len(base_pos) < 40:
        pos = np.array([0.0, -0.2, 0.0]) + rng.uniform(-0.5, 0.5, 3)
        eul = rng.uniform(-np.pi, np.pi, 3)
        pitches = np.linspace(-np.pi, np.pi, 181)
        P = np.tile(pos, (len(pitches), 1))
        E = np.tile(eul, (len(pitches), 1))
        E[:, 1] = pitches
        st = orc.solve_batch(ar, al, P, E, theta_policy=3)["state"]
        idx = [k for k in range(len(pitches) - 1) if {int(st[k]), int(st[k + 1])} == {0, 4}]
        if idx:
            k = idx[0]
            base_pos.append(pos); base_eul.append(eul); lo_hi.append((pitches[k], pitches[k + 1], int(st[k])))
    P, E = [], []
    for pos, eul, (a, b, sa) in zip(base_pos, base_eul, lo_hi):
        for _ in range(60):  # bisection on the checker: the boundary pitch to the last bit
            m = 0.5 * (a + b)
            e = eul.copy(); e[1] = m
            s = int(orc.solve_batch(ar, al, pos[None], e[None], theta_policy=3)["state"][0])
            if s == sa:
                a = m
            else:
                b = m
        for off in 10.0 ** -np.arange(2, 14):
            for sgn in (-1.0, 1.0):
                e = eul.copy(); e[1] = a + sgn * off
                P.append(pos); E.append(e)
                e2 = eul.copy(); e2[1] = b + sgn * off
                P.append(pos); E.append(e2)
    P, E = np.array(P), np.array(E)
    res = to_np(r.solve_batch(soa(P, E, torch_mod)))
    ref = orc.solve_batch(ar, al, P, E)
    np.testing.assert_array_equal(res["reachable"], ref["reachable"])
    np.testing.assert_array_equal(res["state"], ref["state"])
    assert set(np.unique(ref["state"])) >= {0, 4} and 0.2 < ref["reachable"].mean() < 0.8
    m = ref["reachable"].astype(bool)
    # next to tangency the interval ends are ill-conditioned (d angle ~ d disc / (2 sqrt(disc))): 1e-7 there, TOL elsewhere
    width = np.abs(((ref["interval"][:, 1] - ref["interval"][:, 0] + np.pi) % (2 * np.pi)) - np.pi)
    tight = m & (width < 1e-3)
    assert np.max(np.abs(res["interval"][m & ~tight] - ref["interval"][m & ~tight])) < TOL
    if tight.any():
        assert np.max(np.abs(res["interval"][tight] - ref["interval"][tight])) < 1e-7


def test_integration_md_snippet_runs(torch_mod, capsys):
    """INTEGRATION.md section 1 lists the batched entry points as a code block; scripts/integration_snippet.py is that block with
    inputs around it.  It must run as written, and the FK of what solve_batch returns must reproduce the poses."""
    import runpy

    ns = runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "integration_snippet.py"))
    err, ok = ns["err"], ns["ok"]
    assert bool(ok.any()) and float(err[ok].abs().max()) < 1e-12
    assert tuple(ns["out"]["joints"].shape) == (40, 64, 7)
