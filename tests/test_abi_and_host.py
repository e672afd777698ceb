"""CPU-side tests: the C-ABI library loads and exports every symbol include/rsik.h declares (no compute
calls without a GPU), and the host logic (constants, URDF, limits, packing, error behaviour)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from reachy2_symbolic_ik_amd import _abi, build

    build.build()
    return _abi.load()


def test_header_symbols_exported(lib):
    from reachy2_symbolic_ik_amd import _abi

    hdr = open(os.path.join(ROOT, "include", "rsik.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rsik_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/rsik.h but not exported"
    assert declared == set(_abi.PROTOTYPES), "python prototypes out of sync with include/rsik.h"


def test_header_constants_match_python(lib):
    from reachy2_symbolic_ik_amd import _abi, constants

    hdr = open(os.path.join(ROOT, "include", "rsik.h")).read()
    enum = dict((k, int(v)) for k, v in re.findall(r"(RSIK_C_[A-Z_]+|RSIK_ARM_CONSTS_COUNT)\s*=\s*(\d+)", hdr))
    assert enum["RSIK_ARM_CONSTS_COUNT"] == constants.ARM_CONSTS_COUNT == lib.rsik_arm_consts_count()
    for py, c in (("C_SHOULDER", "RSIK_C_SHOULDER"), ("C_TIPL", "RSIK_C_TIPL"), ("C_MST", "RSIK_C_MST"), ("C_TSH", "RSIK_C_TSH"),
                  ("C_ES", "RSIK_C_ES"), ("C_SIDE", "RSIK_C_SIDE"), ("C_PLANE_P", "RSIK_C_PLANE_P"),
                  ("C_PROJ_RADIUS", "RSIK_C_PROJ_RADIUS"), ("C_TIP_Z", "RSIK_C_TIP_Z"), ("C_WRIST_AX", "RSIK_C_WRIST_AX")):
        assert getattr(constants, py) == enum[c]
    states = dict((int(v), k) for k, v in re.findall(r"#define (RSIK_STATE_[A-Z_]+) (\d+)", hdr))
    assert len(states) == 11 and len(constants.STATE_STRINGS) == 11
    assert states[_abi.STATE_EMERGENCY] == "RSIK_STATE_EMERGENCY" and states[_abi.STATE_NOT_REACHABLE_NO_LIMITS] == "RSIK_STATE_NOT_REACHABLE_NO_LIMITS"
    assert int(re.search(r"#define RSIK_SOLVER_STATE_STRIDE (\d+)", hdr).group(1)) == _abi.SOLVER_STATE_STRIDE
    assert lib.rsik_abi_version() == int(re.search(r"#define RSIK_ABI_VERSION (\d+)", hdr).group(1))


def test_no_gpu_fails_loudly(lib):
    """Without a GPU the product must refuse to run rather than fall back to any CPU path."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert lib.rsik_device_count() == 0
    h = C.c_void_p()
    assert lib.rsik_create(0, C.byref(h)) == -2  # RSIK_E_NO_DEVICE
    assert b"no HIP device" in lib.rsik_last_error(None)
    from reachy2_symbolic_ik_amd import ControlIK, SymbolicIK

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SymbolicIK()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ControlIK(urdf_path="config_files/reachy2_ik_minimal.urdf")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "reachy2_symbolic_ik_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower(), f"{f} mentions the checker: the product must not depend on it"


def test_packed_constants_against_reference_goldens(golden_dir):
    from reachy2_symbolic_ik_amd import constants as K

    g = np.load(os.path.join(golden_dir, "g0_constants.npz"))
    for arm in ("r_arm", "l_arm"):
        for tag, so in (("dflt", 0.03), ("ctrl", -1.01)):
            geo = K.ArmGeometry(arm, K.default_ik_parameters(), singularity_offset=so)
            c = geo.pack()
            assert c.shape == (K.ARM_CONSTS_COUNT,)
            p = f"{arm}_{tag}_"
            np.testing.assert_array_equal(c[K.C_SHOULDER:K.C_SHOULDER + 3], g[p + "shoulder_position"])
            assert c[K.C_MAX_LEN] == g[p + "max_arm_length"] and c[K.C_MIN_DIST] == g[p + "shoulder_wrist_min_distance"]
            assert c[K.C_BACKWARD] == g[p + "backward_limit"] and c[K.C_PROJ_MARGIN] == g[p + "projection_margin"]
            np.testing.assert_allclose(c[K.C_ES:K.C_ES + 3], g[p + "elbow_singularity_position"], rtol=0, atol=1e-16)
            np.testing.assert_allclose(geo.wrist_singularity_position, g[p + "wrist_singularity_position"], rtol=0, atol=1e-16)
            assert c[K.C_SIDE] == (1.0 if arm == "r_arm" else -1.0)
            assert c[K.C_ELBOW_LIMIT] == np.radians(127) and c[K.C_SING_OFFSET] == so
            # M_shoulder_torso is a rotation, P_shoulder_torso = -M s
            MsT = c[K.C_MST:K.C_MST + 9].reshape(3, 3)
            np.testing.assert_allclose(MsT @ MsT.T, np.eye(3), atol=1e-15)
            np.testing.assert_allclose(c[K.C_TSH:K.C_TSH + 3], -MsT @ geo.shoulder_position, atol=1e-17)
            # projection plane: unit normal, centre on the plane, radius NaN exactly when the plane misses the sphere (Q18)
            v3 = c[K.C_PLANE_N:K.C_PLANE_N + 3]
            assert abs(np.linalg.norm(v3) - 1) < 1e-15
            assert abs(np.dot(c[K.C_PROJ_CENTER:K.C_PROJ_CENTER + 3] - c[K.C_PLANE_P:K.C_PLANE_P + 3], v3)) < 1e-15
            assert np.isnan(c[K.C_PROJ_RADIUS]) == (so == -1.01)
    with pytest.raises(ValueError, match="arm should be either 'r_arm' or 'l_arm'"):
        K.ArmGeometry("x_arm", K.default_ik_parameters())


def test_urdf_parameters(golden_dir):
    from reachy2_symbolic_ik_amd import constants as K

    g = np.load(os.path.join(golden_dir, "g0_constants.npz"))
    urdf = open(os.path.join(ROOT, "reachy2_symbolic_ik_amd", "config_files", "reachy2_ik_minimal.urdf")).read()
    p = K.get_ik_parameters_from_urdf(urdf, ["r", "l"])
    for arm in ("r_arm", "l_arm"):
        a = arm[0]
        np.testing.assert_array_equal(p[f"{a}_shoulder_position"], g[f"{arm}_urdf_shoulder_position"])
        np.testing.assert_array_equal(p[f"{a}_shoulder_orientation"], g[f"{arm}_urdf_shoulder_orientation_offset"])
        assert p[f"{a}_upper_arm_size"] == g[f"{arm}_urdf_upper_arm_size"] and p[f"{a}_forearm_size"] == g[f"{arm}_urdf_forearm_size"]
        np.testing.assert_array_equal(p[f"{a}_tip_position"], g[f"{arm}_urdf_tip_position"])
    assert K.get_ik_parameters_from_urdf(urdf, []) == {}
    assert set(K.get_ik_parameters_from_urdf(urdf, ["r"])) == {k for k in p if k.startswith("r_")}


def test_interval_limits():
    from reachy2_symbolic_ik_amd.constants import interval_limit_for

    lim, pref = interval_limit_for("r_arm", "unconstrained", -4 * np.pi / 6)
    np.testing.assert_array_equal(lim, [3 * np.pi / 4, -2 * np.pi / 6]) and pref == -4 * np.pi / 6
    lim, pref = interval_limit_for("l_arm", "unconstrained", -4 * np.pi / 6)
    np.testing.assert_allclose(lim, [-2 * np.pi / 3, np.pi / 4], atol=1e-15)  # SURVEY a-16
    assert pref == -np.pi - (-4 * np.pi / 6)
    lim, _ = interval_limit_for("l_arm", "low_elbow", 0.0)
    np.testing.assert_allclose(lim, [-np.pi, -np.pi / 5], atol=1e-15)
    lim, _ = interval_limit_for("r_arm", "low_elbow", 0.0)
    np.testing.assert_array_equal(lim, [-4 * np.pi / 5, 0])


def test_control_constructor_errors():
    """control_ik.py:86-112 error behaviour is host logic and must not need a GPU."""
    from reachy2_symbolic_ik_amd import ControlIK

    with pytest.raises(ValueError, match="No URDF provided"):
        ControlIK()
    with pytest.raises(ValueError, match="Empty URDF file"):
        ControlIK(urdf_path="config_files/does_not_exist.urdf")
    with pytest.raises(ValueError, match="Unknown Reachy model bogus"):
        ControlIK(urdf_path="config_files/reachy2_ik_minimal.urdf", reachy_model="bogus")
    with pytest.raises(ValueError, match="Error while parsing URDF"):
        ControlIK(urdf="<robot><joint></robot>")
    c = ControlIK(urdf_path="config_files/reachy2_ik_minimal.urdf", reachy_model="mini")  # no arms: no GPU needed
    assert c.symbolic_ik_solver == {} and c.nb_search_points == 20 and c.singularity_offset == -1.01


def test_sqrt_threshold_is_exact():
    """RSIK_C_MAX_LEN_SQ: x > T  <=>  sqrt(x) > L for every double x (sqrt correctly rounded, monotonic), checked on
    the doubles around T for a spread of limits including the reference's default max_arm_length."""
    import math

    from reachy2_symbolic_ik_amd.constants import (ArmGeometry, C_MAX_LEN, C_MAX_LEN_SQ, default_ik_parameters,
                                                   sqrt_threshold)

    rng = np.random.default_rng(7)
    limits = [0.65, 0.5, 1.0, 0.1, 2.0 ** -3, 3.0] + list(rng.uniform(0.05, 3.0, size=200))
    for L in limits:
        T = sqrt_threshold(L)
        x = T
        for _ in range(6):  # T and the doubles below it stay inside
            assert not (math.sqrt(x) > L)
            x = math.nextafter(x, 0.0)
        x = T
        for _ in range(6):  # every double above T is outside
            x = math.nextafter(x, math.inf)
            assert math.sqrt(x) > L
    c = ArmGeometry("r_arm", default_ik_parameters()).pack()
    assert c[C_MAX_LEN_SQ] == sqrt_threshold(c[C_MAX_LEN])


def test_packing_helpers():
    import torch

    from reachy2_symbolic_ik_amd.control_ik import matrices_to_m12_soa
    from reachy2_symbolic_ik_amd.symbolic_ik import poses_to_soa

    rng = np.random.default_rng(0)
    M = rng.normal(size=(5, 4, 4))
    m12 = matrices_to_m12_soa(M, torch.device("cpu"))
    assert m12.shape == (12, 5) and m12.is_contiguous()
    np.testing.assert_array_equal(m12[:9].numpy().T.reshape(5, 3, 3), M[:, :3, :3])
    np.testing.assert_array_equal(m12[9:].numpy().T, M[:, :3, 3])
    assert matrices_to_m12_soa(M[0], torch.device("cpu")).shape == (12, 1)
    poses = rng.normal(size=(7, 2, 3))
    s = poses_to_soa(poses, torch.device("cpu"))
    assert s.shape == (6, 7)
    np.testing.assert_array_equal(s.numpy()[:3].T, poses[:, 0])
    np.testing.assert_array_equal(s.numpy()[3:].T, poses[:, 1])
    with pytest.raises(ValueError):
        poses_to_soa(np.zeros((3, 5)), torch.device("cpu"))


def test_utils_format_helpers():
    """README.md:96-110 builds its goal matrix with these two helpers.  The first is packing; the second — like every helper of the
    utils module that computes anything — runs on the device (rsik_matrix_to_pose; tests/test_gpu_utils.py checks its numbers): without
    a GPU it fails loudly instead of answering from the host.  The module itself imports without one, and carries the reference's
    names (utils.py:12-694, less the matplotlib helpers and dead code)."""
    import torch
    from scipy.spatial.transform import Rotation as R

    import reachy2_symbolic_ik_amd.utils as U

    rot = R.from_euler("xyz", [0.3, -0.7, 1.1]).as_matrix()
    M = U.make_homogenous_matrix_from_rotation_matrix([0.55, -0.3, -0.15], rot)
    assert M.shape == (4, 4) and np.array_equal(M[3], [0, 0, 0, 1]) and np.array_equal(M[:3, :3], rot)
    for name in ("rotation_matrix_from_vector", "get_euler_from_homogeneous_matrix", "limit_theta_to_interval", "get_best_discrete_theta",
                 "is_elbow_ok", "is_valid_angle", "angle_diff", "allow_multiturn", "limit_orbita3d_joints", "limit_orbita3d_joints_wrist",
                 "multiturn_safety_check", "continuity_check", "get_ik_parameters_from_urdf", "parse_vector"):
        assert callable(getattr(U, name)), name
    if not torch.cuda.is_available():
        for call in (lambda: U.get_euler_from_homogeneous_matrix(M), lambda: U.angle_diff(1.0, 2.0), lambda: U.rotation_matrix_from_vector(np.ones(3))):
            with pytest.raises(Exception):
                call()
        U.set_default_solver(None)


def test_header_is_plain_c_and_links(tmp_path):
    """include/rsik.h is the drop-in boundary: it must compile as C99 and as C++ on its own, and a C program that
    only knows the header must link against librsik_hip.so and run the calls that need no GPU."""
    import shutil
    import subprocess

    from reachy2_symbolic_ik_amd import _abi

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = os.path.join(root, "include")
    src = tmp_path / "use_rsik.c"
    src.write_text(
        '#include "rsik.h"\n#include <stdio.h>\n'
        "int main(void) {\n"
        "    rsik_ctx *ctx = 0;\n"
        "    if (rsik_abi_version() != RSIK_ABI_VERSION) return 1;\n"
        "    if (rsik_arm_consts_count() != RSIK_ARM_CONSTS_COUNT) return 2;\n"
        "    int rc = rsik_create(rsik_device_count() > 0 ? 0 : 1 << 20, &ctx);\n"
        '    printf("devices=%d create=%d msg=%s\\n", rsik_device_count(), rc, rsik_last_error(ctx));\n'
        "    if (rsik_device_count() <= 0 && rc != RSIK_E_NO_DEVICE) return 3;\n"
        "    if (ctx) rsik_destroy(ctx);\n"
        "    return 0;\n}\n")
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", inc, "-fsyntax-only", str(src)])
    gxx = shutil.which("g++")
    subprocess.check_call([gxx, "-std=c++17", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", "-x", "c++", str(src)])
    _abi.load()  # built
    libdir = os.path.dirname(_abi.LIB_PATH)
    exe = tmp_path / "use_rsik"
    subprocess.check_call([gcc, "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lrsik_hip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md is where a maintainer of the reference looks for what each C entry point replaces: every function
    include/rsik.h declares is named there, and the ABI version it quotes is the header's."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "rsik.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    names = set(re.findall(r"\b(rsik_[a-z0-9_]+)\s*\(", header))
    assert len(names) >= 30
    # (families are written `rsik_malloc / rsik_free / rsik_memcpy_*` and `rsik_control_continuous_step` / `_run` there)
    missing = [n for n in sorted(names)
               if n not in doc and not any(n.startswith(stem[:-1]) for stem in re.findall(r"rsik_[a-z0-9_]+_\*", doc))
               and not (n.endswith("_run") and n[: -len("_run")] + "_step` / `_run" in doc)]
    assert not missing, missing
    version = int(re.search(r"#define\s+RSIK_ABI_VERSION\s+(\d+)", header).group(1))
    assert f"ABI version ({version})" in doc


def test_bench_default_protocol_times_every_gpu_leg_before_any_cpu_leg():
    """bench.py's default run (N = 1, no --config): every config's workload is made resident first, then the GPU legs back to
    back — headline first — and only then the CPU-baseline legs, during which the GPU idles and clocks down (round 4 timed
    configs 3 / 4 / 5 behind a 3.5 s CPU leg).  run_default_protocol with stub legs: the order it drives them in, what it
    reports, and that a failing secondary config costs its own entry only."""
    import bench

    events = []

    def make_leg(cfg):
        def leg():
            events.append(("setup", cfg))
            yield "ready"
            events.append(("gpu", cfg))
            if cfg == 4:
                raise RuntimeError("no such luck")
            yield "timed"
            events.append(("cpu", cfg))
            r = {"kernel_ms": 0.01, "algorithmic_bytes_per_pose": 154, "achieved": 1.0, "frac": 0.1, "traffic": 2.0,
                 "frac_at_286_bytes_state_round_trip_per_step": 0.2}
            return {"config": {"workload": f"config{cfg}"}, "metric": "m", "value": 1.0, "unit": "u", "steps": 20, "warmup": 5, "ms_per_step": 0.4,
                    "launch": "eager", "roofline": r, "cpu_baseline": {"value": 1.0, "parity_on_sample": {"rows": 1}}, "steady_state": {"ms_per_step": 0.39}}
        return leg()

    line, others, order = bench.run_default_protocol(make_leg, 2)
    assert [e for e in events if e[0] == "setup"] == [("setup", 2), ("setup", 3), ("setup", 4), ("setup", 5)]
    phases = [e[0] for e in events]
    assert phases.index("gpu") > max(i for i, p in enumerate(phases) if p == "setup")       # nothing is timed before everything is resident
    assert min(i for i, p in enumerate(phases) if p == "cpu") > max(i for i, p in enumerate(phases) if p == "gpu")  # no CPU leg between GPU legs
    assert [e for e in events if e[0] == "gpu"] == [("gpu", 2), ("gpu", 3), ("gpu", 4), ("gpu", 5)]
    assert [e for e in events if e[0] == "cpu"] == [("cpu", 2), ("cpu", 3), ("cpu", 5)]
    assert line["config"]["workload"] == "config2" and set(others) == {3, 4, 5}
    assert "RuntimeError" in others[4]["error"] and others[3]["kernel_ms"] == 0.01
    # config 5: a bench step is a PASS; kernel time and traffic carried in both units, labelled
    five = others[5]
    assert "kernel_ms" not in five and "traffic" not in five
    assert five["ms_per_pass"] == 0.4 and five["kernel_ms_per_pass"] == 10.0 and five["kernel_ms_per_control_step_of_4096_trajectories"] == 0.01
    assert five["traffic_bytes_per_pass"] == 2000.0 and five["traffic_bytes_per_control_step_of_4096_trajectories"] == 2.0
    assert [o[:2] for o in order] == [["setup", 2], ["setup", 3], ["setup", 4], ["setup", 5], ["gpu", 2], ["gpu", 3], ["gpu", 4], ["gpu", 5],
                                      ["cpu", 2], ["cpu", 3], ["cpu", 5]]


def test_bench_defaults():
    """bench.py's defaults, without a GPU: N > 1 times the north star's job shape (K sharded steps + ONE all-gather) unless told otherwise,
    config 5 the pipelined launch form, the settled leg of configs 2-4 takes 50 ms of untimed launches."""
    import bench

    a = bench.parse_args([])
    assert a.gather == "final" and a.launch == "auto" and a.settle_ms == 50.0 and a.gpus == 1 and a.chunks == 4
    assert bench.parse_args(["--gather-every-step"]).gather == "step" and bench.parse_args(["--graph"]).launch == "graph"
    h = bench.host_facts()
    assert h["cores_visible"] >= 1 and h["logical_cpus"] >= h["cores_visible"] and isinstance(h["cpu_model"], str) and h["cpu_model"]
