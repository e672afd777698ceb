"""GPU tests of the multi-process / multi-GPU path and of the host-side guards added around the C ABI.

On the 1-GPU box the two-rank runs use torch.distributed's gloo backend with both ranks on GPU 0 (`bench.py --backend
gloo --single-device`): every line of the sharding, stripe-pipelined all-gather and rank-0 verification code runs, only
the transport differs.  The RCCL (backend "nccl") variants run when the box has two GPUs and are skipped otherwise.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def _check_two_rank_line(d, gather, world=2):
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["config"]["workload"].startswith("config4")
    m = d["multi_gpu"]
    assert m["kernel_only_solves_per_s"] > 0
    if gather == "none":
        assert m["gather_only_ms"] is None
    else:
        assert m["gather_only_ms"] > 0 and m["end_to_end_solves_per_s"] > 0 and m["gathered_rows_checked"]
        # rank 0 re-solved rows taken from BOTH ranks' parts of the gathered array with the CPU checker
        par = d["cpu_baseline"]["parity_on_sample"]
        assert par["rows"] >= world * 1024 and par["max_abs_joint_error_rad"] < 1e-6
    assert abs(m["end_to_end_solves_per_s"] - d["value"]) < 1e-6 * d["value"]
    if gather == "final":
        # the N > 1 default, the north star's job shape: K sharded steps + ONE all-gather of the final arrays inside the timed region
        assert "ONE RCCL all-gather" in d["config"]["collective"] and "north star" in m["value_is"]
        assert m["gather_final"]["from"].startswith("the timed region") and abs(m["gather_final"]["solves_per_s"] - d["value"]) < 1e-6 * d["value"]
        # (round 6) the other job shape is MEASURED in the same run, in its own overlapped form: a second timed leg, not arithmetic on legs
        assert m["gather_step"]["overlapped"] is True and m["gather_step"]["from"].startswith("a second timed leg")
        assert m["gather_step"]["solves_per_s"] < m["kernel_only_solves_per_s"]
    if gather == "step":
        assert m["gather_step"]["from"].startswith("the timed region") and m["gather_step"]["overlapped"] is True
        assert m["gather_final"]["from"].startswith("a second timed leg")
    if gather != "none":
        # `value` says which job shape it is and since when (the default changed in round 5), both shapes carry their own efficiency
        vd = m["value_definition"]
        assert vd["value_is"] == "gather_" + gather and vd["changed_in_round"] == 5 and d["value_definition"] == vd
        for shape in ("gather_final", "gather_step"):
            assert 0 < m[shape]["efficiency_vs_n1_same_config"] <= 1.0 + 1e-9 and d[shape]["solves_per_s"] == m[shape]["solves_per_s"]
    if gather != "none":  # kernel-only, gather-only and both end-to-end figures are all there, whichever was timed
        assert m["kernel_only_solves_per_s"] > 0 and m["gather_only_solves_per_s"] > 0
        assert m["gather_final"]["solves_per_s"] > 0 and m["gather_step"]["solves_per_s"] > 0
    # what the collective library saw, and the like-for-like reference for the driver's scaling curve
    g = m["group"]
    assert g["world_size"] == world and len(g["ranks"]) == world and {r["rank"] for r in g["ranks"]} == set(range(world))
    assert all(r["pid"] > 0 and r["name"] for r in g["ranks"]) and g["collective_timeout_s"] > 0
    assert m["n1_same_config"]["solves_per_s"] > 0
    eff = m["scaling_efficiency_vs_n1_same_config"]
    assert eff["kernel_only"] is None and eff["kernel_only_note"] and 0 < eff["end_to_end"] <= 1.0 + 1e-9
    # beside `value` at the top level of the line: the efficiency against one GPU on this config, and the north star's job shape
    # (K sharded steps + ONE all-gather of the final joints) end to end
    assert d["scaling_efficiency_vs_n1_same_config"] == eff
    if gather != "none":
        assert m["gather_final"]["solves_per_s"] > 0 and m["gather_step"]["solves_per_s"] > 0
        # (no ordering between the two: under --gather final the first is the timed region itself, the second a separate leg —
        # over gloo, on one GPU, the legs' transport times differ by multiples)
        assert m["gather_final"]["ms_for_K_steps_plus_one_gather"] > 0 and m["gather_final"]["one_all_gather_ms"] > 0
        top = d["gather_final"]
        assert top["solves_per_s"] == m["gather_final"]["solves_per_s"] and 0 < top["efficiency_vs_n1_same_config"] <= 1.0 + 1e-9
        assert d["cpu_baseline"]["workload_filter"]["rows_the_checker_calls_reachable"] == d["cpu_baseline"]["workload_filter"]["rows"]
        assert d["cpu_baseline"]["reference_numpy"]["static"] is True
    else:
        assert "gather_final" not in d


@pytest.mark.parametrize("gather", ["default", "step", "none"])
def test_bench_two_ranks_gloo_one_gpu(gather):
    """`bench.py --gpus 2` starts its own ranks; config 4 (mixed r/l) is the N > 1 default, and so is `--gather final`: the
    all-gather the north star names, once, after the K sharded steps."""
    flags = () if gather == "default" else ("--gather", gather)
    d = _bench("--gpus", "2", "--backend", "gloo", "--single-device", "--poses", "32768", "--steps", "3", "--warmup", "1",
               "--cpu-seconds", "2", "--chunks", "4", *flags)
    _check_two_rank_line(d, "final" if gather == "default" else gather)


@pytest.mark.parametrize("gather", ["final", "step"])
def test_bench_one_rank_rccl_group(gather):
    """`bench.py --gpus 1 --grouped`: the N > 1 code path of the bench on a ONE-rank RCCL group (backend "nccl": what the driver's SCALE runs
    use, and all a one-GPU box can run of it — RCCL refuses two ranks on a device): init_process_group with device_id, describe_group's
    object all-gather, barriers, the in-place stripe all-gathers of both job shapes, the max-over-ranks all-reduce of the timing, the
    checksum exchange, destroy.  With one rank nothing crosses a link; every call is made."""
    d = _bench("--gpus", "1", "--grouped", "--poses", "65536", "--steps", "3", "--warmup", "1", "--cpu-seconds", "2", "--chunks", "4", "--gather", gather)
    assert d["n_gpus"] == 1 and d["config"]["workload"].startswith("config4")
    m = d["multi_gpu"]
    g = m["group"]
    assert g["backend"] == "nccl" and g["world_size"] == 1 and g["collective_library"].startswith("RCCL") and "rehearsal" not in d["config"]["collective"]
    assert m["gather_only_ms"] > 0 and m["gather_final"]["solves_per_s"] > 0 and m["gather_step"]["solves_per_s"] > 0
    assert m["value_definition"]["value_is"] == "gather_" + gather and m["gathered_rows_checked"]
    assert m["xgmi"]["links_usable"] == 0 and m["xgmi"]["frac"] is None
    par = d["cpu_baseline"]["parity_on_sample"]
    assert par["flags_and_states"] == "bit-exact" and par["max_abs_joint_error_rad"] < 1e-9


@pytest.mark.parametrize("gather", ["final", "step"])
def test_bench_four_ranks_gloo_one_gpu(gather):
    """The driver's SCALE command shape rehearsed at the largest rank count a one-GPU box of this pool admits beside the test runner
    (its process guard stops a seventh process on the card; the eight-rank partition itself is covered on the CPU,
    tests/test_distributed_gloo.py; six ranks at BASELINE's 1 048 576 poses per rank are on record under profiles/r06/): four ranks,
    16 384 mixed r/l poses each (gloo moves the gathered arrays through host memory: minutes at full size, and on a busy host even
    32 768 poses in eight stripes took 160 s once), four stripes per shard in the every-step form — the
    rank-0 parity sample is drawn from all four parts of the gathered array, the checksum of every rank's rows is checked on every
    rank, the group as the collective library sees it has four members."""
    d = _bench("--gpus", "4", "--backend", "gloo", "--single-device", "--poses", "16384", "--steps", "2", "--warmup", "1",
               "--cpu-seconds", "2", "--chunks", "4", "--gather", gather)
    _check_two_rank_line(d, gather, world=4)
    assert d["config"]["poses_per_gpu"] == 16384 and "rehearsal" in d["config"]["collective"]


def test_bench_extras_confirm_the_hbm_peak_with_a_copy():
    """the vendor's 8 TB/s is confirmed by a stream copy on the box and the fraction reported against both.  A small config-2
    batch with the extras on: `extras.hbm_copy` holds the copy and read-modify-write rates of a 1 GiB tensor, `roofline` the fraction of
    the better of the two beside the fraction of 8 TB/s."""
    d = _bench("--config", "2", "--poses", "131072", "--steps", "5", "--warmup", "2", "--cpu-seconds", "1", "--settle-ms", "0",
               "--no-live-traffic", "--no-valu-calibration")
    hc = d["extras"]["hbm_copy"]
    assert "error" not in hc and 1000.0 < hc["copy_GBs"] < 8000.0 and 1000.0 < hc["read_modify_write_GBs"] < 8000.0
    r = d["roofline"]
    assert r["peak"] == 8000.0 and r["peak_measured_copy"] == max(hc["copy_GBs"], hc["read_modify_write_GBs"])
    assert abs(r["frac_of_measured_copy"] - r["achieved"] / r["peak_measured_copy"]) < 1e-12 and r["frac_of_measured_copy"] > r["frac"]


def test_bench_measures_traffic_in_its_own_run():
    """roofline.traffic comes from THIS run, not from a file the builder committed: after the timed legs bench.py runs itself twice more
    under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, --kernel-trace only) and reads the kernel's counters.  A
    small config-3 batch: the figure is there, says where it came from, and is the algorithmic 154 B per pose to within a few percent
    (tables, constants and partial lines on a batch this small)."""
    d = _bench("--config", "3", "--poses", "65536", "--steps", "3", "--warmup", "1", "--no-extras", "--cpu-seconds", "1", "--settle-ms", "0")
    assert "steady_state" not in d  # (--settle-ms 0: no settled leg)
    r = d["roofline"]
    assert "traffic_live_error" not in r, r.get("traffic_live_error")
    assert r["traffic_source"].startswith("measured in this run") and r["traffic_seconds"] > 0
    assert 0.98 < r["traffic_over_algorithmic"] < 1.15 and abs(r["traffic"] - r["traffic_over_algorithmic"] * 154 * 65536) < 1.0


def test_bench_two_ranks_gloo_config3_and_config5():
    d = _bench("--gpus", "2", "--backend", "gloo", "--single-device", "--config", "3", "--poses", "16384", "--steps", "2",
               "--warmup", "1", "--cpu-seconds", "2", "--chunks", "2")
    assert d["n_gpus"] == 2 and d["cpu_baseline"]["parity_on_sample"]["max_abs_joint_error_rad"] < 1e-6
    d = _bench("--gpus", "2", "--backend", "gloo", "--single-device", "--config", "5", "--poses", "256", "--steps", "1", "--warmup", "1")
    assert d["n_gpus"] == 2 and d["unit"] == "steps/s" and d["config"]["collective"] == "none"


def test_bench_config5_line_carries_both_protocols():
    """Config 5 on one GPU: the line's value is the protocol asked for (W warm-up passes, K timed) in the form it names — by default
    the K passes issued launch by launch with consecutive passes overlapping (RSIK_OPT_CONT_GOALS_RESIDENT: every pass after the first
    really was issued that way, `run_forms_seen`); `steady_state` holds the same K passes after 60 more untimed ones in all three
    launch forms and one pass on its own (`isolated_pass_ms`); the CPU leg — parity of the LAST timed pass's rows against the
    checker — carries the static reference figures and the host's facts."""
    d = _bench("--config", "5", "--poses", "256", "--steps", "3", "--warmup", "2", "--cpu-seconds", "2")
    assert d["unit"] == "steps/s" and d["steps"] == 3 and d["warmup"] == 2 and d["launch"].startswith("pipelined")
    assert "overlapping" in d["value_is"] and d["run_forms_seen"].get("phased, overlapping the run before", 0) >= 3 + 2 + 60
    ss = d["steady_state"]
    assert ss["launch"] == "pipelined" and ss["steps"] == 3 and ss["after_untimed_passes"] == 65
    f = ss["launch_forms_ms"]
    assert f["pipelined"] > 0 and f["eager"] > 0 and f["graph"] > 0 and ss["value"] > 0 and ss["isolated_pass_ms"] > 0
    cb = d["cpu_baseline"]
    assert cb["parity_on_sample"]["flags_and_states"] == "bit-exact"
    assert cb["reference_numpy"]["config5_steps_per_s_per_core"] > 0
    assert cb["cores_visible"] >= cb["threads_used"] == cb["cores"] >= 1 and cb["cpu_model"] and cb["logical_cpus"] >= cb["cores_visible"]
    g = _bench("--config", "5", "--poses", "256", "--steps", "2", "--warmup", "1", "--launch", "eager", "--no-cpu-baseline", "--no-live-traffic")
    assert g["launch"].startswith("eager") and g["steady_state"]["launch"] == "eager" and g["steady_state"]["launch_forms_ms"]["graph"] > 0
    assert g["steady_state"]["launch_forms_ms"]["pipelined"] > 0 and "without overlap" in g["value_is"]


def test_bench_settled_leg_of_configs_3_and_4():
    """Configs 2-4 on one GPU carry a `steady_state` leg: the same K launches timed again behind >= --settle-ms of untimed launches of the
    same kind, so that the driver's own W = 5 / K = 20 run witnesses the figure the kernel settles to (round 5: the driver read 0.325 for
    config 3 where profiles/ and a longer run read 0.35); the headline figure stays what the protocol says.  And the host's facts sit in
    every CPU leg: the affinity count, the CPU model, the threads the figure was measured with."""
    for cfg, n in ((3, 65536), (4, 131072)):
        d = _bench("--config", str(cfg), "--poses", str(n), "--steps", "4", "--warmup", "2", "--settle-ms", "5", "--no-extras", "--no-live-traffic",
                   "--cpu-seconds", "1")
        ss = d["steady_state"]
        assert ss["steps"] == 4 and ss["untimed_ms_target"] == 5 and ss["untimed_rounds_of_K"] >= 1 and ss["after_untimed_launches"] >= 2 + 4 * 2
        assert ss["ms_per_step"] > 0 and ss["unit"] == "solves/s" and abs(ss["value"] - n / (ss["ms_per_step"] * 1e-3)) < 1e-3 * ss["value"]
        assert abs(ss["frac"] - ss["achieved_GBs"] / 8000.0) < 1e-12 and ss["launch"] in ("graph", "eager")
        assert d["steps"] == 4 and d["warmup"] == 2  # (the headline is the protocol's)
        cb = d["cpu_baseline"]
        assert cb["cores_visible"] >= cb["threads_used"] == cb["cores"] >= 1 and isinstance(cb["cpu_model"], str) and cb["cpu_model"]
        assert str(cb["threads_used"]) in cb["threads_probed"]


def test_stage_timers_script():
    """SURVEY f-4: `scripts/stage_timers.py` (what `bench.py --stages` embeds) — the launch-difference figures of config 2 with the
    product library, and, where a current -DRSIK_TIMELINE_PROBE build travels with the tree, the per-wave stage times of configs 2 / 3."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "stage_timers.py"), "--no-build"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    k = d["launches_config2"]["kernel_us"]
    assert 0 < k["reach_only"] <= k["reach_and_interval"] * 1.1 and k["reach_and_interval"] < k["full_solve"]
    w = d["waves"]
    if "error" not in w:
        assert w["config2"]["stages"] == ["head", "goal", "reach", "joints", "stores"] and min(w["config2"]["mean_us"]) > 0
        assert w["config3"]["stages"] == ["head", "reach", "search", "joints", "safety", "stores"] and min(w["config3"]["mean_us"]) > 0


def test_bench_rank_without_a_device_fails_the_launcher():
    """`--gpus 2` on a box with one GPU and no --single-device: rank 1 has no device of its own and fails; the launcher stops
    the other rank, exits non-zero and shows the failing rank's stderr (never a silent hang, never a shared device: ranks
    that do end up on one device are refused by describe_group unless --single-device says it is a rehearsal)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("on a multi-GPU box every rank gets its own device")
    # one visible GPU: LOCAL_RANK 1 has no device of its own -> the rank fails at set_device; the launcher must fail too
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--poses", "4096", "--steps", "1",
                        "--warmup", "1", "--no-cpu-baseline", "--no-live-traffic"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert "rank" in p.stderr and "exited with code" in p.stderr


def test_bench_two_ranks_rccl():
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    d = _bench("--gpus", "2", "--poses", "262144", "--steps", "5", "--warmup", "2", "--cpu-seconds", "2")
    _check_two_rank_line(d, "step")
    assert "rehearsal" not in d["config"]["collective"] and d["multi_gpu"]["xgmi"]["achieved"] > 0
    g = d["multi_gpu"]["group"]
    assert g["distinct_devices"] == 2 and g["collective_library"].startswith("RCCL") and g["backend"] == "nccl"


def _rccl_worker(rank, world, port, result_dir):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from oracle import oracle as orc
        from reachy2_symbolic_ik_amd import DualArmIK
        from reachy2_symbolic_ik_amd.distributed import solve_sharded

        import contextlib
        import io

        rng = np.random.default_rng(5)
        n = 100003  # not a multiple of world * chunks: the last stripe is padded
        pos = np.array([0.25, 0.0, -0.15]) + rng.uniform(-0.3, 0.3, size=(n, 3))
        eul = np.array([0, -np.pi / 2, 0]) + rng.uniform(-0.8, 0.8, size=(n, 3))
        arm = (rng.uniform(size=n) < 0.5).astype(np.uint8)
        pos[:, 1] += np.where(arm == 1, 0.2, -0.2)
        with contextlib.redirect_stdout(io.StringIO()):
            dual = DualArmIK(device=rank)
        cols = torch.as_tensor(np.ascontiguousarray(np.concatenate([pos.T, eul.T], axis=0))).cuda()
        arm_t = torch.as_tensor(arm).cuda()
        lo_of = {}

        def solve_fn(c, out):
            lo = (c.data_ptr() - cols.data_ptr()) // 8  # the column slice's first row
            lo_of[lo] = True
            dual.solve_batch(arm_t[lo: lo + c.shape[1]], c, want_elbow=False, out=out)

        ref = orc.solve_batch(orc.Arm("r_arm", 0.03), orc.Arm("l_arm", 0.03), pos, eul, arm_id=arm, nthreads=8)
        for chunks in (1, 4):
            full = solve_sharded(solve_fn, cols, gather=("joints", "state"), chunks=chunks)
            torch.cuda.synchronize()
            np.testing.assert_array_equal(full["state"].cpu().numpy(), ref["state"])
            ok = ref["reachable"].astype(bool)
            assert np.max(np.abs(full["joints"].cpu().numpy()[ok] - ref["joints"][ok])) < 1e-9
        open(os.path.join(result_dir, f"ok_{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_solve_sharded_rccl_two_gpus(tmp_path):
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rccl_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(tmp_path / f"ok_{r}") for r in range(2))


def test_solve_sharded_rccl_one_rank(tmp_path):
    """The same worker as the two-GPU test on a ONE-rank RCCL group (what a one-GPU box allows: RCCL refuses two ranks on a device):
    torch's "nccl" backend initialised with device_id, the in-place all_gather_into_tensor of a stripe (input = this rank's rows of the
    output), async work handles, chunks = 1 and 4 with a padded last stripe, destroy — every RCCL call of the sharded path except the
    transport between devices."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rccl_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(tmp_path / "ok_0")


def test_c_abi_allgather_single_rank():
    """rsik_comm_* / rsik_allgather (include/rsik.h): a one-rank communicator on GPU 0 — unique id, init, the all-gather
    (out of place and in place) on the context's stream, destroy.  With one rank the collective is a copy, which is
    enough to show that librccl is found, the by-value ncclUniqueId call is laid out correctly and the stream is used."""
    import torch

    from reachy2_symbolic_ik_amd import HipSolver

    hs = HipSolver(0)
    L = hs.lib
    uid = (C.c_char * 128)()
    assert L.rsik_comm_unique_id(C.cast(uid, C.c_void_p)) == 0
    comm = C.c_void_p()
    hs._check(L.rsik_comm_init_rank(hs._h, 1, 0, C.cast(uid, C.c_void_p), C.byref(comm)))
    assert comm.value
    src = torch.arange(7 * 1000, dtype=torch.float64, device="cuda").reshape(1000, 7)
    dst = torch.zeros_like(src)
    hs._bind_stream()
    hs._check(L.rsik_allgather(hs._h, comm, src.data_ptr(), dst.data_ptr(), src.numel() * 8))
    hs._check(L.rsik_allgather(hs._h, comm, src.data_ptr(), src.data_ptr(), src.numel() * 8))  # in place
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    assert L.rsik_allgather(hs._h, None, src.data_ptr(), dst.data_ptr(), 8) != 0  # NULL communicator is an error, not a crash
    hs._check(L.rsik_comm_destroy(hs._h, comm))


def test_out_buffers_are_validated():
    """Caller-supplied `out` tensors reach the kernels as raw pointers: wrong dtype / shape / device / layout must
    raise instead of corrupting memory."""
    import contextlib
    import io

    import torch

    from reachy2_symbolic_ik_amd import ControlIK, SymbolicIK

    with contextlib.redirect_stdout(io.StringIO()):
        ik = SymbolicIK("r_arm")
        c = ControlIK(urdf_path="config_files/reachy2_ik_minimal.urdf")
    n = 64
    poses = torch.zeros((6, n), dtype=torch.float64, device="cuda")
    good = lambda: {"joints": torch.empty((n, 7), dtype=torch.float64, device="cuda")}  # noqa: E731
    ik.solve_batch(poses, out=good())
    bad = [
        {"joints": torch.empty((n, 7), dtype=torch.float32, device="cuda")},
        {"joints": torch.empty((n - 1, 7), dtype=torch.float64, device="cuda")},
        {"joints": torch.empty((n, 7), dtype=torch.float64)},
        {"joints": torch.empty((n, 14), dtype=torch.float64, device="cuda")[:, ::2]},
        {"reachable": torch.empty((n,), dtype=torch.int32, device="cuda")},
        {"interval": torch.empty((2, n), dtype=torch.float64, device="cuda").t()},
    ]
    for o in bad:
        with pytest.raises(ValueError):
            ik.solve_batch(poses, out=o)
    M = np.tile(np.eye(4), (n, 1, 1))
    for o in bad[:4] + [{"emergency": torch.empty((n,), dtype=torch.float64, device="cuda")}]:
        with pytest.raises(ValueError):
            c.symbolic_inverse_kinematics_batch("r_arm", M, out=o)
    st = c.new_continuous_state("r_arm", n)
    with pytest.raises(ValueError):
        c.symbolic_inverse_kinematics_continuous_batch("r_arm", M, st, out=bad[0])
    with pytest.raises(ValueError):
        c.run_continuous_trajectories("r_arm", np.tile(np.eye(4), (3, n, 1, 1)), st, out={"joints": torch.empty((n, 3, 7), dtype=torch.float64, device="cuda")})
    # a column slice of a larger SoA batch is consumed in place (no copy) and gives the same answer
    big = torch.rand((6, 4 * n), dtype=torch.float64, device="cuda") * 0.2
    a = ik.solve_batch(big[:, n: 2 * n])
    b = ik.solve_batch(big[:, n: 2 * n].contiguous())
    assert torch.equal(a["state"], b["state"]) and torch.equal(a["interval"].nan_to_num(9.0), b["interval"].nan_to_num(9.0))


def test_plan_keeps_its_stream_and_notices_new_constants():
    import contextlib
    import io

    import torch

    from reachy2_symbolic_ik_amd import HipSolver, SymbolicIK

    hs = HipSolver(0)
    with contextlib.redirect_stdout(io.StringIO()):
        ik = SymbolicIK("r_arm", solver=hs)
    rng = np.random.default_rng(3)
    n = 4096
    pos = np.array([0.0, -0.2, 0.0]) + rng.uniform(-0.5, 0.5, size=(n, 3))
    poses = torch.as_tensor(np.ascontiguousarray(np.concatenate([pos.T, rng.uniform(-3, 3, size=(n, 3)).T]))).cuda()
    out = {"state": torch.full((n,), 255, dtype=torch.uint8, device="cuda")}
    plan = ik.solve_batch(poses, out=out, plan_only=True)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):  # another call leaves the context bound to a side stream ...
        ik.solve_batch(poses)
    plan["launch"]()               # ... the plan still goes to the stream it was planned on
    torch.cuda.current_stream().synchronize()
    ref = ik.solve_batch(poses)["state"]
    torch.cuda.synchronize()
    assert torch.equal(out["state"], ref)
    with contextlib.redirect_stdout(io.StringIO()):
        other = SymbolicIK("r_arm", singularity_offset=-1.01, solver=hs)  # same arm slot, other constants
    other.solve_batch(poses)
    with pytest.raises(RuntimeError):
        plan["launch"]()


def test_options_are_validated_and_build_id_matches_sources():
    from reachy2_symbolic_ik_amd import HipSolver, _abi, build

    hs = HipSolver(0)
    for opt, bad in ((_abi.OPT_SWEEP_MODE, 3), (_abi.OPT_NO_TIPZ, 2), (_abi.OPT_CONT_RUN_MODE, 5), (99, 0), (_abi.OPT_SWEEP_MODE, -1)):
        with pytest.raises(_abi.RsikError):
            hs.set_option(opt, bad)
    hs.set_option(_abi.OPT_SWEEP_MODE, 2)
    assert hs.get_option(_abi.OPT_SWEEP_MODE) == 2
    assert hs.build_id() == "RSIK_SRC_HASH=" + build.source_hash()  # the library in use was built from the sources next to it


def test_clock_monitor_reports_a_plausible_clock():
    import torch

    from reachy2_symbolic_ik_amd import HipSolver

    hs = HipSolver(0)
    side = torch.cuda.Stream()
    core, real = hs.clock_monitor(0.02, 8, side)
    a = torch.rand(1 << 20, dtype=torch.float64, device="cuda")
    for _ in range(20):
        hs.debug_math(6, a, a)
    torch.cuda.synchronize()
    ghz = (core / real * 0.1).cpu().numpy()
    assert np.all(real.cpu().numpy() >= 0.02e8 * 0.99) and np.all((ghz > 0.3) & (ghz < 2.6)), ghz
