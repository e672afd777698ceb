"""GPU tests of what round 6 added to rsik_control_continuous_run (run with -m gpu on an MI355X):

* RSIK_OPT_CONT_GOALS_RESIDENT — consecutive runs of one shape whose prepare phase starts beside the tail of the run before
  (include/rsik.h): whatever the overlap, every output and the trajectory state must be what the same runs give one after the
  other (the reference carries its state from call to call, control_ik.py:80-83, 276-407: nothing of a call may depend on when the
  next one is issued);
* rsik_control_continuous_last_form — how a run was issued, in particular the launch-per-step fallback for a solver whose
  projection margin lets is_reachable_no_limits fail (symbolic_ik.py:343-345, control_ik.py:385-387).
"""
import contextlib
import io

import numpy as np
import pytest

from test_gpu_parity import _abi_mod, _same_run, make_control, torch_mod  # noqa: F401

pytestmark = pytest.mark.gpu


def _outs(torch, n_steps, n_traj):
    return {"joints": torch.empty((n_steps, n_traj, 7), dtype=torch.float64, device="cuda"),
            "reachable": torch.empty((n_steps, n_traj), dtype=torch.uint8, device="cuda"),
            "state": torch.empty((n_steps, n_traj), dtype=torch.uint8, device="cuda")}


def _eventful(torch, traj, every=5, at=0.4):
    """Goals that jump part-way for every `every`-th trajectory: the continuity check trips and the trajectory stays latched
    (control_ik.py:205-210, 398-405) — chunks that go through the chain phase's step-by-step path, fill_rest, emergency states
    written over the prepare phase's rows."""
    t = traj.clone()
    n_steps = t.shape[0]
    k = int(n_steps * at)
    sel = torch.arange(0, t.shape[2], every, device=t.device)
    t[k:, 9, sel] -= 0.25  # x of the goal position
    t[k:, 11, sel] += 0.2  # z
    return t


def _snapshot(res, st):
    got = {k: v.clone() for k, v in res.items()}
    got["cont_state"] = st[:11].clone()
    return got


@pytest.mark.parametrize("n_traj,n_steps,block", [(777, 208, 0), (300, 208, 16), (64, 96, 32), (1030, 400, 0)])
@pytest.mark.parametrize("buffers", [1, 2])
def test_overlapping_runs_are_the_serial_runs(torch_mod, n_traj, n_steps, block, buffers):
    """K runs issued back to back with the promise set — the bench's protocol (the state is reset and every trajectory
    re-initialises, same goals) and a stream of different goals continuing one state — into one set of output buffers (each block's
    reachable / state rows then wait for the previous run's chain kernel of the same rows) or two taking turns; blocks of 16 steps
    make 13 blocks, more than the eight workspace slots, so that slots go round within a run and from run to run.  Against the same
    runs with the promise not set."""
    from bench import make_config5_trajectories

    torch = torch_mod
    A = _abi_mod()
    K = 5
    # one long eventful trajectory batch: the "continue" protocol walks it chunk by chunk, the "reset" protocol repeats its first chunk
    whole = _eventful(torch, make_config5_trajectories(n_traj, n_steps * K, seed=600 + n_traj), every=5, at=0.3 / K + 0.5)
    goals = [whole[k * n_steps: (k + 1) * n_steps].contiguous() for k in range(K)]
    goals[0] = _eventful(torch, goals[0], every=7, at=0.6)
    c = make_control()
    hs = c._solver
    hs.set_option(A.OPT_CONT_BLOCK_STEPS, block)
    st0 = c.new_continuous_state("r_arm", n_traj)
    for protocol in ("reset", "continue"):
        results = {}
        for resident in (False, True):
            st = st0.clone()
            outs = [_outs(torch, n_steps, n_traj) for _ in range(buffers)]
            forms, snaps = [], []
            for k in range(K):
                if protocol == "reset":
                    st.copy_(st0)
                res = c.run_continuous_trajectories("r_arm", goals[k if protocol == "continue" else 0], st,
                                                    first_step_timed_out=(protocol == "reset" or k == 0), current_pose=goals[0][0],
                                                    out=outs[k % buffers], goals_resident=resident)
                forms.append(res.run_form)
                if buffers == 2:  # (the other set is the next run's: reading this one between the calls is within the promise)
                    snaps.append(_snapshot(res, st))
            torch.cuda.synchronize()
            snaps.append(_snapshot(outs[(K - 1) % buffers], st))
            results[resident] = snaps
            want = A.CONT_FORM_PHASED_OVERLAPPED if resident else A.CONT_FORM_PHASED
            # (the first run with the promise may itself overlap the last one without: with eight or more blocks the two use the same slots)
            assert (resident or forms[0] == A.CONT_FORM_PHASED) and all(f == want for f in forms[1:]), (protocol, resident, forms)
        assert len(results[True]) == len(results[False])
        latched = int((results[False][-1]["cont_state"][9] != 0).sum())
        assert latched > 0, "the test's trajectories were meant to trip the continuity check"
        for k, (a, b) in enumerate(zip(results[False], results[True])):
            _same_run(torch, a, b, (protocol, buffers, k), joint_tol=0.0)  # the same kernels on the same data: bit for bit
    hs.set_option(A.OPT_CONT_BLOCK_STEPS, 0)


def test_overlap_only_behind_a_run_of_the_same_shape(torch_mod):
    """The promise takes effect behind a run of the same n and block length on the same stream into the same workspace; anything
    else is issued as without it, and says so: another number of trajectories, another block length, another stream, a context a
    hipGraph points into, a launch-per-step run in between.  Results are right either way."""
    from bench import make_config5_trajectories

    torch = torch_mod
    A = _abi_mod()
    c = make_control()
    hs = c._solver
    hs.set_option(A.OPT_CONT_GOALS_RESIDENT, 1)
    ref_c = make_control()

    def run(n_traj, n_steps, seed, ctrl=c, **kw):
        traj = make_config5_trajectories(n_traj, n_steps, seed=seed)
        st = ctrl.new_continuous_state("r_arm", n_traj)
        res = ctrl.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0], **kw)
        return res, st

    def check(n_traj, n_steps, seed, want_form, tol=0.0, **kw):
        res, st = run(n_traj, n_steps, seed, **kw)
        assert res.run_form == want_form, (res.run_form_name, want_form)
        ref, st_ref = run(n_traj, n_steps, seed, ctrl=ref_c)
        torch.cuda.synchronize()
        _same_run(torch, _snapshot(ref, st_ref), _snapshot(res, st), (n_traj, n_steps, seed), joint_tol=tol)

    check(500, 192, 1, A.CONT_FORM_PHASED)                 # the first run of the context
    check(500, 192, 2, A.CONT_FORM_PHASED_OVERLAPPED)
    check(500, 96, 3, A.CONT_FORM_PHASED_OVERLAPPED)       # fewer steps, the same block length (64): still the same slots
    check(500, 384, 3, A.CONT_FORM_PHASED)                 # another block length (a third of the run: 128)
    check(500, 384, 4, A.CONT_FORM_PHASED_OVERLAPPED)
    check(321, 96, 5, A.CONT_FORM_PHASED)                  # another n
    check(321, 96, 6, A.CONT_FORM_PHASED_OVERLAPPED)
    with torch.cuda.stream(torch.cuda.Stream()):           # another stream: it waits for the run before (cont_run_begin)
        check(321, 96, 7, A.CONT_FORM_PHASED)
        check(321, 96, 8, A.CONT_FORM_PHASED_OVERLAPPED)
        torch.cuda.current_stream().synchronize()
    check(321, 96, 9, A.CONT_FORM_PHASED)
    hs.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_STEPS)   # a launch per step in between writes outputs on the caller's stream
    check(321, 96, 10, A.CONT_FORM_STEPS, tol=1e-9)        # (step kernel against pipeline: _same_run's bar)
    hs.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_AUTO)
    check(321, 96, 11, A.CONT_FORM_PHASED)
    check(321, 96, 12, A.CONT_FORM_PHASED_OVERLAPPED)
    check(321, 96, 13, A.CONT_FORM_PHASED, goals_resident=False)   # the promise withdrawn for one call: that run keeps its own slots only,
    check(321, 96, 14, A.CONT_FORM_PHASED)                         # so the next one meets it first
    check(321, 96, 15, A.CONT_FORM_PHASED_OVERLAPPED)
    # a hipGraph recorded from this context points into its workspace: the library cannot see replays, so nothing overlaps any more
    traj = make_config5_trajectories(321, 96, seed=15)
    st = c.new_continuous_state("r_arm", 321)
    graph, out = c.capture_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0])
    assert hs.continuous_last_form() == A.CONT_FORM_PHASED_CAPTURED
    graph.replay()
    torch.cuda.synchronize()
    check(321, 96, 16, A.CONT_FORM_PHASED)
    check(321, 96, 17, A.CONT_FORM_PHASED)
    hs.set_option(A.OPT_CONT_GOALS_RESIDENT, 0)


def test_streamed_chunks_of_a_long_trajectory(torch_mod):
    """What the promise is for: a long trajectory batch handed over in chunks whose goals are resident ahead of time — run k + 1
    carries on from run k's state (no re-initialisation), its prepare phase beside run k's tail.  The chunks' outputs together
    are the one long run's, and so is the state at the end (flags, states and the carried theta bit for bit; joints to 1e-9: the
    cut into blocks differs, _same_run)."""
    from bench import make_config5_trajectories

    torch = torch_mod
    n_traj, chunk, n_chunks = 900, 160, 6
    traj = _eventful(torch, make_config5_trajectories(n_traj, chunk * n_chunks, seed=77), every=7, at=0.55)
    c = make_control()
    st_long = c.new_continuous_state("r_arm", n_traj)
    long_run = c.run_continuous_trajectories("r_arm", traj, st_long, first_step_timed_out=True, current_pose=traj[0])
    torch.cuda.synchronize()
    st = c.new_continuous_state("r_arm", n_traj)
    pieces = [traj[k * chunk: (k + 1) * chunk].contiguous() for k in range(n_chunks)]
    torch.cuda.synchronize()  # every chunk's goals are on the device before the first run is issued: the promise
    outs, forms = [], []
    for k in range(n_chunks):
        res = c.run_continuous_trajectories("r_arm", pieces[k], st, first_step_timed_out=(k == 0), current_pose=traj[0], goals_resident=True)
        outs.append(res)
        forms.append(res.run_form)
    torch.cuda.synchronize()
    A = _abi_mod()
    assert forms[0] == A.CONT_FORM_PHASED and all(f == A.CONT_FORM_PHASED_OVERLAPPED for f in forms[1:]), forms
    got = {k: torch.cat([o[k] for o in outs], dim=0) for k in ("joints", "reachable", "state")}
    got["cont_state"] = st[:11].clone()
    ref = _snapshot(long_run, st_long)
    assert int((ref["cont_state"][9] != 0).sum()) > 0
    _same_run(torch, ref, got, "chunks")


def test_last_form_says_when_a_run_fell_back_to_a_launch_per_step(torch_mod):
    """symbolic_ik.py:343-345: the pulled-back wrist of is_reachable_no_limits lands inside u + f only for a positive
    projection_margin; with 0 (or less) it can fail, the reference raises on purpose (control_ik.py:385-387) and the pipeline's
    phases do not carry that outcome: such a solver's runs are n_steps launches of the step kernel whatever RSIK_OPT_CONT_RUN_MODE
    says.  Correct — the same bits as asking for RSIK_CONT_RUN_STEPS — and visible: rsik_control_continuous_last_form /
    the result's `run_form`."""
    from bench import make_config5_trajectories
    from reachy2_symbolic_ik_amd import SymbolicIK

    torch = torch_mod
    A = _abi_mod()
    n_traj, n_steps = 200, 64
    traj = make_config5_trajectories(n_traj, n_steps, seed=5150)
    c = make_control()
    hs = c._solver
    assert hs.continuous_last_form() == A.CONT_FORM_NONE
    st = c.new_continuous_state("r_arm", n_traj)
    res = c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0])
    assert res.run_form == A.CONT_FORM_PHASED and res.run_form_name == "phased"
    with contextlib.redirect_stdout(io.StringIO()):
        c.symbolic_ik_solver["r_arm"] = SymbolicIK("r_arm", projection_margin=0.0, singularity_offset=-1.01,
                                                   wrist_limit=np.rad2deg(c.orbita3D_max_angle), solver=c._solver)
    runs = {}
    for mode in (A.CONT_RUN_AUTO, A.CONT_RUN_PHASED, A.CONT_RUN_STEPS):
        hs.set_option(A.OPT_CONT_RUN_MODE, mode)
        st = c.new_continuous_state("r_arm", n_traj)
        res = c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0], goals_resident=True)
        torch.cuda.synchronize()
        want = A.CONT_FORM_STEPS if mode == A.CONT_RUN_STEPS else A.CONT_FORM_STEPS_NO_LIMITS_CAN_FAIL
        assert res.run_form == want == hs.continuous_last_form(), (mode, res.run_form_name)
        assert "projection margin" in res.run_form_name or mode == A.CONT_RUN_STEPS
        runs[mode] = _snapshot(res, st)
    hs.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_AUTO)
    # the same launches in all three: bit for bit, the NaN rows of the steps whose goal is_reachable_no_limits refused included
    ref = runs[A.CONT_RUN_STEPS]
    assert int((ref["state"] == A.STATE_NOT_REACHABLE_NO_LIMITS).sum()) > 0, "a zero margin was meant to let is_reachable_no_limits fail"
    for mode in (A.CONT_RUN_AUTO, A.CONT_RUN_PHASED):
        for k in ref:
            assert torch.equal(ref[k].contiguous().view(torch.uint8), runs[mode][k].contiguous().view(torch.uint8)), (mode, k)


def test_run_numbers_start_over_before_they_wrap(torch_mod, tmp_path):
    """The stream-ordering words of rsik_control_continuous_run hold a 32-bit run number and every wait is "word >= number": the library
    drains and starts them over before the number wraps (4e9 runs — weeks of a control loop issuing a run per tick).  A build with that
    point at run 5 (-DRSIK_EDGE_SEQ_WRAP=5) against the product library: 14 overlapping runs continuing one eventful trajectory batch,
    every run's outputs and the carried state bit for bit; the runs right behind a restart are issued as first runs (not overlapping)."""
    import os
    import subprocess
    import sys

    torch = torch_mod
    A = _abi_mod()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from reachy2_symbolic_ik_amd import build as B

    lib = os.path.join(root, "build", "variants", "seqwrap.so")

    def stale(path):
        code = "import ctypes as C,sys; l=C.CDLL(sys.argv[1]); l.rsik_build_id.restype=C.c_char_p; print(l.rsik_build_id().decode())"
        p = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True)
        return p.returncode != 0 or B.source_hash() not in p.stdout

    if not os.path.exists(lib) or stale(lib):
        subprocess.run([sys.executable, os.path.join(root, "scripts", "build_variant.py"), "seqwrap", "-DRSIK_EDGE_SEQ_WRAP=5"],
                       check=True, stdout=subprocess.DEVNULL, timeout=900)
    got = {}
    for name, arg in (("product", "-"), ("wraps", lib)):
        out = str(tmp_path / f"{name}.pt")
        p = subprocess.run([sys.executable, os.path.join(root, "scripts", "probes", "seq_wrap_check.py"), arg, out],
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        got[name] = torch.load(out)
    a, b = got["product"], got["wraps"]
    assert a["forms"] == [A.CONT_FORM_PHASED] + [A.CONT_FORM_PHASED_OVERLAPPED] * 13
    # the test build: runs 1-5 carry the numbers 1-5; run 6 finds the number at its limit, starts over and forks like a first run; so do
    # runs 11 (and 16 ...)
    assert b["forms"] == [A.CONT_FORM_PHASED if k in (0, 5, 10) else A.CONT_FORM_PHASED_OVERLAPPED for k in range(14)], b["forms"]
    assert int((a["runs"][-1]["cont_state"][9] != 0).sum()) > 0, "the trajectories were meant to trip the continuity check"
    for k, (ra, rb) in enumerate(zip(a["runs"], b["runs"])):
        for key in ra:
            assert torch.equal(ra[key].contiguous().view(torch.uint8), rb[key].contiguous().view(torch.uint8)), (k, key)
