"""GPU tests of the utils surface (run with -m gpu on an MI355X): the policy layer's helpers as the reference's utils module
exposes them (utils.py:93-112, 334-396, 443-589; callers import them, src/example/test_ik.py:16-21, test_go_to.py:10-13) are
rsik_stage operations here — the reference's own sequence of operations on explicit arguments, on the device — behind functions
of the same names.  Against G18: what the reference's functions returned for the same arguments (oracle/gen_golden.py gen_utils)."""
import os

import numpy as np
import pytest

from test_gpu_parity import _abi_mod, load, make_symbolic, torch_mod  # noqa: F401

pytestmark = pytest.mark.gpu


def test_utils_stages_against_reference(golden_dir, torch_mod):
    """Every helper as a batch through HipSolver.stage: outcomes exact, numbers to 1e-12 (1e-9 for the Orbita3D cone, which goes
    through two Euler conversions)."""
    A = _abi_mod()
    g = load(golden_dir, "g18_utils.npz")
    hs, r, l = make_symbolic(0.03)
    T = lambda *cols: torch_mod.as_tensor(np.ascontiguousarray(np.column_stack(cols), dtype=np.float64)).cuda()  # noqa: E731
    run = lambda op, rows: hs.stage(op, rows, 0).cpu().numpy()  # noqa: E731

    o = run(A.STAGE_ANGLE_DIFF, T(g["ad_a"], g["ad_b"]))
    assert np.max(np.abs(o[:, 0] - g["ad_out"])) < 1e-12
    # (Python's % on the same doubles: the device's modulo is exact, so are most of these to the last bit)
    assert (o[:, 0] == g["ad_out"]).mean() > 0.95
    o = run(A.STAGE_IS_VALID_ANGLE, T(g["iv_angle"], g["iv_interval"]))
    np.testing.assert_array_equal(o[:, 0] != 0, g["iv_valid"] != 0)
    o = run(A.STAGE_LIMIT_THETA_TO_INTERVAL, T(g["iv_angle"], g["iv_prev"], g["iv_interval"]))
    np.testing.assert_array_equal(o[:, 1] != 0, g["lt_inside"] != 0)
    assert np.max(np.abs(o[:, 0] - g["lt_theta"])) < 1e-12
    o = run(A.STAGE_IS_ELBOW_OK, T(g["eo_elbow"], g["eo_side"], g["eo_so"], g["eo_coeff"], g["eo_esp"]))
    np.testing.assert_array_equal(o[:, 0] != 0, g["eo_ok"] != 0)
    o = run(A.STAGE_ALLOW_MULTITURN, T(g["mt_new"], g["mt_prev"]))
    assert np.max(np.abs(o - g["mt_out"])) < 1e-12
    o = run(A.STAGE_MULTITURN_SAFETY_CHECK, T(g["ms_joints"], g["ms_limits"]))
    assert np.max(np.abs(o[:, :7] - g["ms_out"])) == 0.0
    np.testing.assert_array_equal(o[:, 7] != 0, g["ms_stop"] != 0)
    n = len(g["cc_joints"])
    o = run(A.STAGE_CONTINUITY_CHECK, T(g["cc_joints"], g["cc_prev"], np.tile(g["cc_max"], (n, 1))))
    np.testing.assert_array_equal(o[:, 7] != 0, g["cc_stop"] != 0)
    assert np.max(np.abs(o[:, :7] - g["cc_out"])) == 0.0
    o = run(A.STAGE_LIMIT_ORBITA3D_JOINTS, T(g["lo_joints"], g["lo_max"]))
    # (angles are compared as rotations of a circle: -pi and pi are the same answer)
    d = np.abs((o - g["lo_out"] + np.pi) % (2 * np.pi) - np.pi)
    assert d.max() < 1e-9, float(d.max())
    a = g["bd_args"]
    o = run(A.STAGE_BEST_DISCRETE_THETA, T(a[:, 0], g["bd_interval"], a[:, 1:], g["bd_circle"]))
    np.testing.assert_array_equal(o[:, 0] != 0, g["bd_found"] != 0)
    np.testing.assert_array_equal(o[:, 2] != 0, g["bd_worked"] != 0)
    assert np.max(np.abs(o[:, 1] - g["bd_theta"])) < 1e-12  # (a grid point, the preferred theta, or previous_theta handed back)
    # a grid size that is not a sane number is an empty grid (nothing found, previous_theta handed back), never a loop count
    pick = np.nonzero(np.abs(g["bd_interval"]).sum(axis=1) < 6.0)[0][:4]  # (not the whole circle: NaN is in no other interval)
    bad = np.column_stack([a[pick, 0], g["bd_interval"][pick], a[pick, 1:], g["bd_circle"][pick]])
    bad[:, 3] = [np.nan, -5.0, 1e300, np.inf]
    bad[:, 4] = np.nan  # (a preferred theta that is in no interval: the shortcut cannot answer)
    o = run(A.STAGE_BEST_DISCRETE_THETA, torch_mod.as_tensor(np.ascontiguousarray(bad)).cuda())
    assert np.all(o[:, 0] == 0) and np.array_equal(o[:, 1], bad[:, 0])
    with pytest.raises(Exception):
        hs.stage(A.STAGE_ANGLE_DIFF, T(g["ad_a"]), 0)  # a row of the wrong length is refused on the host


def test_nearest_approach_of_nearly_parallel_planes(golden_dir, torch_mod):
    """points_of_nearest_approach (symbolic_ik.py:588-606, np.linalg.lstsq at :580) and the circle-line points behind it for planes
    whose normals are 1e-6 ... 1e-3 apart — just outside normal_vector_margin, where the 3 x 2 system's condition number is 1 / delta.
    The stage solves it by QR (round 5's normal equations squared the condition: 1e-4 relative at 1e-6, the advisor's finding): q agrees
    with the reference's SVD to ~1e-14 / delta relative (two backward-stable solves of a system whose condition number is 1 / delta: 3e-9
    at 1e-6, where the normal equations gave 1e-4), the found / not-found decision (Q7) exactly."""
    A = _abi_mod()
    g = load(golden_dir, "g18_utils.npz")
    hs, r, l = make_symbolic(0.03)
    rows = g["np_in"]
    na = hs.stage(A.STAGE_NEAREST_APPROACH, torch_mod.as_tensor(np.ascontiguousarray(rows[:, :12])).cuda(), 0).cpu().numpy()
    np.testing.assert_array_equal(na[:, 0] != 0, g["np_found"] != 0)
    assert np.max(np.abs(na[:, 4:7] - g["np_v"])) < 1e-9
    found = g["np_found"] != 0
    rel = np.linalg.norm(na[found, 1:4] - g["np_q"][found], axis=1) / np.maximum(1.0, np.linalg.norm(g["np_q"][found], axis=1))
    bound = 2e-14 / g["np_delta"][found]
    assert np.all(rel < np.maximum(bound, 1e-12)), (float(rel.max()), float((rel / bound).max()))
    cl_in = np.concatenate([rows[found, 0:3], rows[found, 12:13], g["np_v"][found], g["np_q"][found]], axis=1)
    cl = hs.stage(A.STAGE_CIRCLE_LINE, torch_mod.as_tensor(np.ascontiguousarray(cl_in)).cuda(), 0).cpu().numpy()
    np.testing.assert_array_equal(cl[:, 0].astype(np.uint8), g["np_cl_count"][found])
    want = g["np_cl_points"][found]
    m = ~np.isnan(want)
    assert np.array_equal(np.isnan(cl[:, 1:]), ~m) and np.max(np.abs(cl[:, 1:][m] - want[m]), initial=0.0) < 1e-9


def test_utils_module_is_the_reference_surface(golden_dir, torch_mod):
    """The module-level functions: names, arguments, return types and texts of the reference's utils (a sample of G18 through each,
    one call at a time), on a device context the module creates itself — importing it needs no GPU, calling it does."""
    import reachy2_symbolic_ik_amd.utils as U
    from scipy.spatial.transform import Rotation as R

    g = load(golden_dir, "g18_utils.npz")
    U.set_default_solver(0)
    for k in range(0, 400, 37):
        assert abs(U.angle_diff(float(g["ad_a"][k]), float(g["ad_b"][k])) - g["ad_out"][k]) < 1e-12
    for k in range(0, 500, 23):
        iv = g["iv_interval"][k]
        assert U.is_valid_angle(float(g["iv_angle"][k]), iv) == bool(g["iv_valid"][k])
        theta, text = U.limit_theta_to_interval(float(g["iv_angle"][k]), float(g["iv_prev"][k]), iv)
        assert abs(float(theta) - g["lt_theta"][k]) < 1e-12 and text == ("theta in interval" if g["lt_inside"][k] else "theta not in interval")
    for k in range(0, 600, 41):
        assert U.is_elbow_ok(g["eo_elbow"][k], int(g["eo_side"][k]), float(g["eo_so"][k]), float(g["eo_coeff"][k]), g["eo_esp"][k]) == bool(g["eo_ok"][k])
    for k in range(0, 400, 29):
        out = U.allow_multiturn(list(g["mt_new"][k]), list(g["mt_prev"][k]), "r_arm")
        assert isinstance(out, list) and np.max(np.abs(np.array(out) - g["mt_out"][k])) < 1e-12
        j, stop, text = U.multiturn_safety_check(list(g["ms_joints"][k]), *[float(v) for v in g["ms_limits"][k]], "before")
        assert np.array_equal(np.array(j), g["ms_out"][k]) and stop == bool(g["ms_stop"][k]) and text == str(g["ms_text"][k])
        j, stop, text = U.continuity_check(np.array(g["cc_joints"][k]), np.array(g["cc_prev"][k]), list(g["cc_max"]), "")
        assert np.array_equal(np.asarray(j), g["cc_out"][k]) and stop == bool(g["cc_stop"][k]) and text == str(g["cc_text"][k])
    for k in range(0, 500, 31):
        out = U.limit_orbita3d_joints(list(g["lo_joints"][k]), float(g["lo_max"][k]))
        d = np.abs((np.array(out) - g["lo_out"][k] + np.pi) % (2 * np.pi) - np.pi)
        assert isinstance(out, list) and d.max() < 1e-9
    for k in range(0, 40, 7):
        out = U.limit_orbita3d_joints_wrist(list(g["low_joints"][k]), float(np.deg2rad(42.5)))
        d = np.abs((np.array(out) - g["low_out"][k] + np.pi) % (2 * np.pi) - np.pi)
        assert d.max() < 1e-9 and np.array_equal(np.array(out)[:4], g["low_joints"][k][:4])
    # get_best_discrete_theta the way ControlIK calls it (control_ik.py:424-434): on a solver that has just answered is_reachable
    hs, r, l = make_symbolic(0.03)
    pose = np.array([[0.38, -0.2, -0.28], [0.0, -np.pi / 2, 0.0]])
    ok, interval, fn, _ = r.is_reachable(pose)
    assert ok
    found, theta, text = U.get_best_discrete_theta(0.1, interval, r.get_elbow_position, 20, -4 * np.pi / 6, "r_arm", r.singularity_offset,
                                                   r.singularity_limit_coeff, r.elbow_singularity_position)
    assert text.startswith("r_arm\ninterval: ") and isinstance(found, bool)
    if found:
        assert U.is_elbow_ok(r.get_elbow_position(theta), 1, r.singularity_offset, r.singularity_limit_coeff, r.elbow_singularity_position)
    with pytest.raises(TypeError):
        U.get_best_discrete_theta(0.1, interval, lambda t: np.zeros(4), 20, 0.0, "r_arm", 0.03, 1.0, np.zeros(3))
    # the two format helpers of README.md:96-110, the second now through rsik_matrix_to_pose
    rot = R.from_euler("xyz", [0.3, -0.7, 1.1]).as_matrix()
    M = U.make_homogenous_matrix_from_rotation_matrix([0.55, -0.3, -0.15], rot)
    pos, eul = U.get_euler_from_homogeneous_matrix(M)
    np.testing.assert_allclose(pos, [0.55, -0.3, -0.15])
    np.testing.assert_allclose(eul, [0.3, -0.7, 1.1], atol=1e-12)
    assert np.allclose(U.get_euler_from_homogeneous_matrix(M, degrees=True)[1], np.degrees([0.3, -0.7, 1.1]))
    assert np.max(np.abs(U.rotation_matrix_from_vector(np.array([0.2, -0.5, 0.7])) @ np.array([1.0, 0, 0]) - np.array([0.2, -0.5, 0.7]) / np.linalg.norm([0.2, -0.5, 0.7]))) < 1e-12
    assert U.utils_on_device() is not None
    U.set_default_solver(None)


def test_utils_helpers_from_two_threads(torch_mod):
    """The module's one context is shared under a lock: two threads calling helpers at once get their own answers."""
    import threading

    import reachy2_symbolic_ik_amd.utils as U

    rng = np.random.default_rng(3)
    a, b = rng.uniform(-20, 20, size=(2, 200))
    want = ((a - b + np.pi) % (2 * np.pi)) - np.pi
    got = [np.zeros(200), np.zeros(200)]

    def work(t):
        for k in range(200):
            got[t][k] = U.angle_diff(float(a[k]), float(b[k])) if t == 0 else U.angle_diff(float(b[k]), float(a[k]))

    ts = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert np.max(np.abs(got[0] - want)) < 1e-12
    assert np.max(np.abs(got[1] - (((b - a + np.pi) % (2 * np.pi)) - np.pi))) < 1e-12
