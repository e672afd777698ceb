"""GPU tests of include/rsik.h "Rows that are not numbers" and of empty batches, through the C ABI.

A row whose goal holds a NaN / an infinity is answered in that row (RSIK_STATE_INVALID_INPUT, unreachable, NaN outputs), the same
way by the HIP kernels and by the CPU checker; every other row is bit for bit what it is in a clean run — in every form of the
continuous run as well, where a poisoned trajectory must not stall or disturb the chain phase, the theta phase or their hand-overs.
The reference's own behaviour on such input (exceptions, one hang, numbers derived from infinities) is on record in
tests/golden/g13_hostile.npz and pinned against the checker's convention in tests/test_oracle_golden.py.
"""
import numpy as np
import pytest

from test_gpu_parity import URDF, make_control, make_symbolic, orc, soa, to_np, torch_mod  # noqa: F401

pytestmark = pytest.mark.gpu

INVALID = 10
POISONS = (float("nan"), float("inf"), float("-inf"))


def _bits_equal(torch, a, b):
    return torch.equal(a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8))


def _rows_except(torch, n, rows):
    keep = torch.ones(n, dtype=torch.bool, device="cuda")
    keep[torch.as_tensor(rows, device="cuda")] = False
    return keep


def test_solve_rows_that_are_not_numbers(torch_mod, orc, golden_dir):
    """rsik_solve (r/l mixed, every theta policy's output set) and rsik_reach_state: each of the six pose entries poisoned with
    NaN / +inf / -inf in rows scattered over several workgroups, plus the rows of G13 (a)."""
    torch = torch_mod
    solver, r, l = make_symbolic(0.03)
    from reachy2_symbolic_ik_amd import DualArmIK

    rng = np.random.default_rng(5)
    n = 3000
    pos = np.array([0.0, -0.2, 0.0]) + rng.uniform(-0.6, 0.6, size=(n, 3))
    eul = rng.uniform(-np.pi, np.pi, size=(n, 3))
    clean = {k: v.clone() for k, v in r.solve_batch(soa(pos, eul, torch)).items()}
    bad_rows = rng.choice(n, size=18, replace=False)
    p2, e2 = pos.copy(), eul.copy()
    for q, row in enumerate(bad_rows):
        (p2 if q % 6 < 3 else e2)[row, q % 3] = POISONS[q // 6]
    got = r.solve_batch(soa(p2, e2, torch))
    torch.cuda.synchronize()
    keep = _rows_except(torch, n, bad_rows)
    for k in clean:
        assert _bits_equal(torch, got[k][keep], clean[k][keep]), k
    b = torch.as_tensor(bad_rows, device="cuda")
    assert (got["state"][b] == INVALID).all() and (got["reachable"][b] == 0).all()
    for k in ("joints", "interval", "elbow"):
        assert torch.isnan(got[k][b]).all(), k
    # the checker says the same of every row
    ref = orc.solve_batch(orc.Arm("r_arm", 0.03), orc.Arm("l_arm", 0.03), p2, e2)
    np.testing.assert_array_equal(to_np(got)["state"], ref["state"])
    np.testing.assert_array_equal(to_np(got)["reachable"], ref["reachable"])
    # reachability only (theta policy "none") and the scalar drop-in's solver-state kernel
    only = r.is_reachable_batch(soa(p2, e2, torch))
    assert (only["state"][b] == INVALID).all() and _bits_equal(torch, only["state"][keep], clean["state"][keep])
    g = np.load(f"{golden_dir}/g13_hostile.npz")
    dual = DualArmIK(device=0)
    res = to_np(dual.solve_batch(g["sym_arm"], soa(g["sym_pos"], g["sym_eul"], torch)))
    assert (res["state"] == INVALID).all() and (res["reachable"] == 0).all() and np.isnan(res["joints"]).all()
    st = solver.new_solver_state(n)
    rs = solver.reach_state(soa(pos, eul, torch), st)
    st2 = st.clone()
    rs2 = solver.reach_state(soa(p2, e2, torch), st2)
    torch.cuda.synchronize()
    assert (rs2["state"][b] == INVALID).all() and (rs2["reachable"][b] == 0).all()
    assert _bits_equal(torch, rs2["state"][keep], rs["state"][keep]) and _bits_equal(torch, st2[keep], st[keep])
    assert _bits_equal(torch, st2[b][:, :16], st[b][:, :16])  # the solver object of a refused pose keeps its geometry


@pytest.mark.parametrize("sweep_mode", [0, 1, 2])
def test_control_discrete_rows_that_are_not_numbers(torch_mod, orc, golden_dir, sweep_mode):
    """rsik_control_discrete: each of the twelve matrix entries poisoned; the wave-cooperative sweep (mode 1) shares lanes between
    poses, so a poisoned pose must drop out of it.  current_joints that are not numbers: IEEE propagation in that row only."""
    torch = torch_mod
    from bench import make_config3_matrices

    A = __import__("reachy2_symbolic_ik_amd")._abi
    n = 2048
    M = make_config3_matrices(n, seed=11)
    c = make_control()
    c._solver.set_option(A.OPT_SWEEP_MODE, sweep_mode)
    clean = {k: v.clone() for k, v in c.symbolic_inverse_kinematics_batch("r_arm", M).items()}
    rng = np.random.default_rng(6)
    bad_rows = rng.choice(n, size=36, replace=False)
    Mp = M.copy()
    for q, row in enumerate(bad_rows):
        Mp[row, (q % 12) // 4, (q % 12) % 4] = POISONS[q // 12]
    got = c.symbolic_inverse_kinematics_batch("r_arm", Mp)
    torch.cuda.synchronize()
    keep = _rows_except(torch, n, bad_rows)
    b = torch.as_tensor(bad_rows, device="cuda")
    for k in clean:
        assert _bits_equal(torch, got[k][keep], clean[k][keep]), k
    assert (got["state"][b] == INVALID).all() and (got["reachable"][b] == 0).all() and (got["emergency"][b] == 0).all()
    assert torch.isnan(got["joints"][b]).all()
    ar, al = orc.Arm("r_arm", -1.01), orc.Arm("l_arm", -1.01)
    ref = orc.control_discrete_batch(ar, al, Mp, nb_search_points=int(c.nb_search_points))
    np.testing.assert_array_equal(to_np(got)["state"], ref["state"])
    np.testing.assert_array_equal(to_np(got)["reachable"], ref["reachable"])
    # G13 (b) through the drop-in, both arms
    g = np.load(f"{golden_dir}/g13_hostile.npz")
    res = to_np(c.symbolic_inverse_kinematics_batch(torch.as_tensor(g["disc_arm"]).cuda(), g["disc_M"]))
    assert (res["state"] == INVALID).all() and np.isnan(res["joints"]).all()
    # current_joints (what an unreachable row returns, C:457-458) that are not numbers
    cj = np.zeros((n, 7))
    cj[bad_rows[0], 2] = np.nan
    cj[bad_rows[1], 0] = np.inf
    base = c.symbolic_inverse_kinematics_batch("r_arm", M, current_joints=np.zeros((n, 7)))
    base = {k: v.clone() for k, v in base.items()}
    got = c.symbolic_inverse_kinematics_batch("r_arm", M, current_joints=cj)
    torch.cuda.synchronize()
    keep2 = _rows_except(torch, n, bad_rows[:2])
    for k in base:
        assert _bits_equal(torch, got[k][keep2], base[k][keep2]), k
    assert _bits_equal(torch, got["state"], base["state"]) and _bits_equal(torch, got["reachable"], base["reachable"])
    c._solver.set_option(A.OPT_SWEEP_MODE, 0)


def _poisoned_trajectories(torch, traj):
    """The goal matrices of a run [n_steps, 12, n_traj] with goals that are not numbers in seven trajectories: a single step in the
    middle, the first step (which also (re)initialises), the last step, a stretch across a block and chunk boundary, every step
    from some point on, the whole trajectory, and two neighbours in one group of eight."""
    n_steps, _, n = traj.shape
    t = traj.clone()
    nan, inf = float("nan"), float("inf")
    cases = {}
    picks = [3, n // 5, n // 3, n // 2, n // 2 + 1, (2 * n) // 3, n - 1]
    picks = sorted(set(min(max(p, 0), n - 1) for p in picks))
    while len(picks) < 7:  # (tiny batches: fewer distinct trajectories)
        picks.append(picks[-1])
    t[n_steps // 2, 9, picks[0]] = nan
    cases[picks[0]] = [n_steps // 2]
    if n > 1:
        t[0, 0, picks[1]] = inf
        cases.setdefault(picks[1], []).append(0)
        t[n_steps - 1, 4, picks[2]] = -inf
        cases.setdefault(picks[2], []).append(n_steps - 1)
        lo, hi = max(1, n_steps // 3 - 5), min(n_steps, n_steps // 3 + 6)
        t[lo:hi, 11, picks[3]] = nan
        cases.setdefault(picks[3], []).extend(range(lo, hi))
        t[(2 * n_steps) // 3:, 5, picks[4]] = nan
        cases.setdefault(picks[4], []).extend(range((2 * n_steps) // 3, n_steps))
        t[:, 10, picks[5]] = inf
        cases.setdefault(picks[5], []).extend(range(n_steps))
        t[n_steps // 4, 2, picks[6]] = nan
        cases.setdefault(picks[6], []).append(n_steps // 4)
    return t, {k: sorted(set(v)) for k, v in cases.items()}


def _m12_to_matrices(m12):
    """[n_steps, 12, k] -> [n_steps, k, 4, 4]"""
    s, _, k = m12.shape
    Ms = np.tile(np.eye(4), (s, k, 1, 1))
    Ms[:, :, :3, :3] = np.moveaxis(m12[:, :9, :], 1, 2).reshape(s, k, 3, 3)
    Ms[:, :, :3, 3] = np.moveaxis(m12[:, 9:, :], 1, 2)
    return Ms


@pytest.mark.parametrize("n_traj,n_steps", [(300, 200), (9, 131), (1, 64)])
def test_continuous_run_goals_that_are_not_numbers(torch_mod, orc, n_traj, n_steps):
    """rsik_control_continuous_run, every form (the step kernel launch by launch, the phased pipeline with its edge kinds and
    block sizes): poisoned trajectories are reported step by step as the checker reports them, end in the state the checker ends
    in, and every other trajectory — outputs and carried state — is bit for bit that of a clean run of the same form."""
    torch = torch_mod
    from bench import make_config5_trajectories

    A = __import__("reachy2_symbolic_ik_amd")._abi
    traj = make_config5_trajectories(n_traj, n_steps, seed=123)
    bad, cases = _poisoned_trajectories(torch, traj)
    c = make_control()
    hs = c._solver
    st0 = c.new_continuous_state("r_arm", n_traj)
    keep = _rows_except(torch, n_traj, sorted(cases))
    forms = [("steps", A.CONT_RUN_STEPS, 0, 0), ("phased", A.CONT_RUN_PHASED, 0, 0), ("phased/events", A.CONT_RUN_PHASED, 0, A.PHASED_EDGES_BY_EVENT),
             ("phased/blocks of 24", A.CONT_RUN_PHASED, 24, 0), ("phased/no theta-first", A.CONT_RUN_PHASED, 40, A.PHASED_NO_THETA_FIRST)]
    results = {}
    try:
        for name, mode, block, variant in forms:
            hs.set_option(A.OPT_CONT_RUN_MODE, mode)
            hs.set_option(A.OPT_CONT_BLOCK_STEPS, block)
            hs.set_option(A.OPT_CONT_PHASED_VARIANT, variant)
            runs = []
            for T in (traj, bad):
                st = st0.clone()
                o = c.run_continuous_trajectories("r_arm", T, st, first_step_timed_out=True, current_pose=T[0])
                hs.synchronize()
                runs.append(({k: v.clone() for k, v in o.items()}, st.clone()))
            (co, cst), (bo, bst) = runs
            for k in co:
                assert _bits_equal(torch, bo[k][:, keep], co[k][:, keep]), (name, k)
            assert _bits_equal(torch, bst[:, keep], cst[:, keep]), (name, "cont_state")
            results[name] = (bo, bst)
    finally:
        hs.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_AUTO)
        hs.set_option(A.OPT_CONT_BLOCK_STEPS, 0)
        hs.set_option(A.OPT_CONT_PHASED_VARIANT, 0)
    # the forms agree with each other on the poisoned trajectories too: flags, states, theta bit for bit, joints to 1e-9 (NaN where NaN)
    ref_o, ref_st = results["steps"]
    for name, (o, st) in results.items():
        assert _bits_equal(torch, o["state"], ref_o["state"]) and _bits_equal(torch, o["reachable"], ref_o["reachable"]), name
        assert torch.equal(torch.isnan(o["joints"]), torch.isnan(ref_o["joints"])), name
        assert float((torch.nan_to_num(o["joints"]) - torch.nan_to_num(ref_o["joints"])).abs().max()) <= 1e-9, name
        assert _bits_equal(torch, st[0], ref_st[0]) and torch.equal(st[8:11], ref_st[8:11]), name
        assert float((st[1:8] - ref_st[1:8]).abs().max()) <= 1e-9, name
    # ... and with the checker, step by step
    a = orc.Arm("r_arm", -1.01)
    cols = sorted(cases)
    Ms = _m12_to_matrices(bad[:, :, torch.as_tensor(cols).cuda()].cpu().numpy())
    J, F, S = (ref_o[k].cpu().numpy() for k in ("joints", "reachable", "state"))
    for col, k in enumerate(cols):
        cs = orc.ContinuousState(c.previous_theta["r_arm"], c.previous_sol["r_arm"])
        for i in range(n_steps):
            j, ok, code = orc.control_continuous_step(a, cs, Ms[i, col], timed_out=(i == 0), preferred_theta_arg=-4 * np.pi / 6,
                                                      preferred_theta_self=c.preferred_theta["r_arm"], constrained_mode=0,
                                                      current_joints=cs.previous_sol, current_pose=Ms[0, col])
            assert ok == bool(F[i, k]) and code == S[i, k], (k, i, code, S[i, k])
            if i in cases[k] and code != 8:
                assert code == INVALID and np.isnan(J[i, k]).all(), (k, i)
            else:
                assert np.max(np.abs(j - J[i, k])) < 1e-7, (k, i)
        assert abs(cs.previous_theta - float(ref_st[0, k])) < 1e-9 or (np.isnan(cs.previous_theta) and bool(torch.isnan(ref_st[0, k])))
        assert cs.emergency_stop == bool(ref_st[9, k] != 0)


def test_continuous_step_goal_that_is_not_a_number(torch_mod, orc):
    """rsik_control_continuous_step: the poisoned step of a trajectory leaves its carried state as the checker leaves it, and a
    poisoned row of cont_state / current_joints / current_pose stays in its row."""
    torch = torch_mod
    from bench import make_config5_trajectories

    n, steps = 200, 12
    traj = make_config5_trajectories(n, steps, seed=31)
    Ms = _m12_to_matrices(traj.cpu().numpy())           # [steps, n, 4, 4]
    bad = Ms.copy()
    bad[5, 17, 1, 3] = np.nan
    bad[0, 80, 0, 0] = np.inf
    bad[6:9, 150, 2, 2] = np.nan
    c, c2 = make_control(), make_control()
    st, st2 = c.new_continuous_state("r_arm", n), c2.new_continuous_state("r_arm", n)
    a = orc.Arm("r_arm", -1.01)
    rows = [17, 80, 150]
    states = {k: orc.ContinuousState(c.previous_theta["r_arm"], c.previous_sol["r_arm"]) for k in rows}
    keep = _rows_except(torch, n, rows)
    for i in range(steps):
        to = np.full(n, 1 if i == 0 else 0, dtype=np.uint8)
        clean = c.symbolic_inverse_kinematics_continuous_batch("r_arm", Ms[i], st, timed_out=to, current_pose=Ms[0])
        got = c2.symbolic_inverse_kinematics_continuous_batch("r_arm", bad[i], st2, timed_out=to, current_pose=Ms[0])
        torch.cuda.synchronize()
        for k in clean:
            assert _bits_equal(torch, got[k][keep], clean[k][keep]), (i, k)
        assert _bits_equal(torch, st2[:, keep], st[:, keep]), i
        g = to_np(got)
        for k in rows:
            cs = states[k]
            j, ok, code = orc.control_continuous_step(a, cs, bad[i, k], timed_out=(i == 0), preferred_theta_arg=-4 * np.pi / 6,
                                                      preferred_theta_self=c.preferred_theta["r_arm"], constrained_mode=0,
                                                      current_joints=cs.previous_sol, current_pose=Ms[0, k])
            assert ok == bool(g["reachable"][k]) and code == g["state"][k], (i, k)
            assert np.array_equal(np.isnan(j), np.isnan(g["joints"][k])) and np.nanmax(np.abs(j - g["joints"][k]), initial=0.0) < 1e-7
            assert abs(cs.previous_theta - float(st2[0, k])) < 1e-9 and np.max(np.abs(cs.previous_sol - st2[1:8, k].cpu().numpy())) < 1e-7
    # a trajectory state / start-up input that is not a number: its own row only, and the call completes
    st3 = st.clone()
    st3[0, 7] = float("nan")
    st3[3, 90] = float("inf")
    cj = np.zeros((n, 7))
    cj[33, 1] = np.nan
    ref = c.symbolic_inverse_kinematics_continuous_batch("r_arm", Ms[3], st.clone(), current_joints=np.zeros((n, 7)))
    ref = {k: v.clone() for k, v in ref.items()}
    got = c.symbolic_inverse_kinematics_continuous_batch("r_arm", Ms[3], st3, current_joints=cj)
    torch.cuda.synchronize()
    keep = _rows_except(torch, n, [7, 90])
    for k in ref:
        assert _bits_equal(torch, got[k][keep], ref[k][keep]), k
    assert (got["state"] <= INVALID).all()


def test_empty_batches_on_every_entry_point(torch_mod):
    """n = 0 (and n_steps = 0) is a valid call everywhere: RSIK_OK, nothing launched, nothing written."""
    torch = torch_mod
    solver, r, l = make_symbolic(0.03)
    f64 = torch.float64
    z6 = torch.zeros((6, 0), dtype=f64, device="cuda")
    for theta in ("interval0", "none", ("explicit", torch.zeros((0,), dtype=f64, device="cuda"))):
        res = r.solve_batch(z6, theta=theta)
        assert res["state"].shape == (0,) and (theta == "none" or res["joints"].shape == (0, 7))
    st = solver.new_solver_state(0)
    assert solver.reach_state(z6, st)["state"].shape == (0,)
    c = make_control()
    res = c.symbolic_inverse_kinematics_batch("r_arm", np.zeros((0, 4, 4)))
    assert res["joints"].shape == (0, 7) and res["emergency"].shape == (0,)
    cst = c.new_continuous_state("r_arm", 0)
    res = c.symbolic_inverse_kinematics_continuous_batch("r_arm", np.zeros((0, 4, 4)), cst)
    assert res["joints"].shape == (0, 7)
    A = __import__("reachy2_symbolic_ik_amd")._abi
    for mode in (A.CONT_RUN_AUTO, A.CONT_RUN_PHASED, A.CONT_RUN_STEPS):
        c._solver.set_option(A.OPT_CONT_RUN_MODE, mode)
        res = c.run_continuous_trajectories("r_arm", torch.zeros((5, 12, 0), dtype=f64, device="cuda"), cst)
        assert res["joints"].shape == (5, 0, 7)
        cst4 = c.new_continuous_state("r_arm", 4)
        before = cst4.clone()
        res = c.run_continuous_trajectories("r_arm", torch.zeros((0, 12, 4), dtype=f64, device="cuda"), cst4)
        assert res["joints"].shape == (0, 4, 7) and torch.equal(before, cst4)
    c._solver.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_AUTO)
    c._solver.control_continuous_reserve(0, 0)
    pose = solver.matrix_to_pose(torch.zeros((12, 0), dtype=f64, device="cuda")) if hasattr(solver, "matrix_to_pose") else None
    assert pose is None or pose.shape == (6, 0)
    fk = solver.forward_kinematics(torch.zeros((0, 7), dtype=f64, device="cuda"), arm_uniform=0) if hasattr(solver, "forward_kinematics") else None
    assert fk is None or (fk[0] if isinstance(fk, tuple) else fk).shape[0] == 0
    c._solver.synchronize()


def test_scalar_drop_in_raises_like_the_reference(torch_mod):
    """The scalar API keeps the reference's behaviour for a pose that is not numbers: numpy.linalg.LinAlgError (symbolic_ik.py:580),
    the solver object and the ControlIK state as they were."""
    solver, r, l = make_symbolic(0.03)
    good = np.array([[0.55, -0.3, -0.15], [0.0, -np.pi / 2, 0.0]])
    ok, interval, fn, state = r.is_reachable(good)
    assert ok and state == "reachable"
    wrist = r.wrist_position.copy()
    bad = good.copy()
    bad[1, 0] = np.nan
    with pytest.raises(np.linalg.LinAlgError):
        r.is_reachable(bad)
    with pytest.raises(np.linalg.LinAlgError):
        r.is_reachable_no_limits(bad)
    assert np.array_equal(r.wrist_position, wrist)
    j, _ = fn(interval[0])
    assert np.isfinite(j).all()
    c = make_control()
    M = np.eye(4)
    M[:3, 3] = [0.4, -0.25, -0.2]
    j0, ok0, s0 = c.symbolic_inverse_kinematics("r_arm", M, "continuous")
    th, ps = c.previous_theta["r_arm"], np.array(c.previous_sol["r_arm"]).copy()
    Mb = M.copy()
    Mb[1, 3] = np.nan
    for kind in ("discrete", "continuous"):
        with pytest.raises(np.linalg.LinAlgError):
            c.symbolic_inverse_kinematics("r_arm", Mb, kind)
    assert abs(c.previous_theta["r_arm"] - th) < 1e-15 and np.array_equal(np.array(c.previous_sol["r_arm"]), ps)
    j1, ok1, s1 = c.symbolic_inverse_kinematics("r_arm", M, "continuous")
    twin = make_control()  # the same two good calls, nothing in between
    twin.symbolic_inverse_kinematics("r_arm", M, "continuous")
    j2, ok2, s2 = twin.symbolic_inverse_kinematics("r_arm", M, "continuous")
    assert ok1 == ok2 and s1 == s2 and np.max(np.abs(np.array(j1) - np.array(j2))) < 1e-9
    assert abs(c.previous_theta["r_arm"] - twin.previous_theta["r_arm"]) < 1e-12


def test_arrays_past_two_and_four_gib(torch_mod):
    """Maximum sizes: 80 Mi poses / goal matrices through ONE rsik_solve / rsik_control_discrete launch (joints 4.4 GiB, the
    matrices 7.5 GiB) and 1 Mi trajectories x 96 steps through one continuous run (joints 5.3 GiB; the sequential phases address a
    block through 2 GiB buffer windows, cont_plan sizes the blocks for that): the same 1 Mi poses / 4096 trajectories tiled, every
    tile's rows bit-identical to the first tile's and to the same rows solved as a batch of their own — a 32-bit offset anywhere
    would show as a tile that differs or was never written (scripts/probes/large_batches.py)."""
    import os
    import subprocess
    import sys

    free, _total = torch_mod.cuda.mem_get_info()
    if free < 64 * 2**30:
        pytest.skip("needs 64 GiB of free device memory")
    torch_mod.cuda.empty_cache()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "probes", "large_batches.py"), "80", "1048576"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "TOTAL ok" in r.stdout and r.stdout.count("every tile identical to the first") == 12, r.stdout[-3000:]
