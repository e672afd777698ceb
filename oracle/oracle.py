"""ctypes front-end of the CPU checker library (oracle/librsik_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librsik_oracle.so")

STATE_STRINGS = [
    "reachable",
    "Pose out of reach",
    "Backward pose",
    "wrist out of range",
    "limited by wrist",
    "out of reach - should not happen",
    "limited by shoulder",
    "",
]

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_ip = C.POINTER(C.c_int)


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("rsik_oracle.c", "rsik_oracle.h"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None
_native = None


def _bind(L):
    L.orc_arm_ndoubles.restype = C.c_int
    L.orc_solver_ndoubles.restype = C.c_int
    L.orc_max_threads.restype = C.c_int
    L.orc_angle_diff.restype = C.c_double
    L.orc_angle_diff.argtypes = [C.c_double, C.c_double]
    L.orc_pymod.restype = C.c_double
    L.orc_pymod.argtypes = [C.c_double, C.c_double]
    L.orc_limit_theta_to_interval.restype = C.c_double
    L.orc_limit_theta_to_interval.argtypes = [C.c_double, C.c_double, _dp]
    L.orc_is_valid_angle.restype = C.c_int
    L.orc_is_valid_angle.argtypes = [C.c_double, _dp]
    L.orc_get_best_theta_to_current_joints.restype = C.c_double
    L.orc_get_best_theta_to_current_joints.argtypes = [_dp, _dp, _dp, C.c_int, C.c_double]
    L.orc_arm_init.argtypes = [_dp, C.c_int, _dp, _dp, C.c_double, C.c_double, _dp] + [C.c_double] * 7
    L.orc_arm_init_default.argtypes = [_dp, C.c_int, C.c_double]
    L.orc_is_reachable.restype = C.c_int
    L.orc_is_reachable.argtypes = [_dp, _dp, _dp, _dp, _ip, _dp]
    L.orc_is_reachable_no_limits.restype = C.c_int
    L.orc_is_reachable_no_limits.argtypes = [_dp, _dp, _dp, _dp]
    L.orc_get_elbow_position.argtypes = [_dp, C.c_double, _dp]
    L.orc_get_joints.restype = C.c_int
    L.orc_get_joints.argtypes = [_dp, _dp, C.c_double, _dp, _dp, _dp]
    L.orc_limit_orbita3d_joints.argtypes = [_dp, C.c_double, _dp]
    L.orc_rotation_matrix_from_vector.argtypes = [_dp, _dp]
    L.orc_euler_from_matrix_xyz.argtypes = [_dp, _dp]
    L.orc_get_best_discrete_theta.restype = C.c_int
    L.orc_get_best_discrete_theta.argtypes = [_dp, _dp, C.c_double, _dp, C.c_int, C.c_double, _dp]
    L.orc_is_elbow_ok_args.restype = C.c_int
    L.orc_is_elbow_ok_args.argtypes = [_dp, C.c_double, C.c_double, C.c_double, _dp]
    L.orc_allow_multiturn.argtypes = [_dp, _dp, _dp]
    L.orc_multiturn_safety_check.restype = C.c_int
    L.orc_multiturn_safety_check.argtypes = [_dp, _dp, _dp]
    L.orc_continuity_check.restype = C.c_int
    L.orc_continuity_check.argtypes = [_dp, _dp, _dp, _dp]
    L.orc_best_discrete_theta_circle.restype = C.c_int
    L.orc_best_discrete_theta_circle.argtypes = [C.c_double, _dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, _dp, _dp, _dp, _ip]
    L.orc_points_of_nearest_approach.restype = C.c_int
    L.orc_points_of_nearest_approach.argtypes = [_dp] * 6
    L.orc_intersection_circle_line.restype = C.c_int
    L.orc_intersection_circle_line.argtypes = [_dp, C.c_double, _dp, _dp, _dp]
    L.orc_control_discrete.restype = C.c_int
    L.orc_control_discrete.argtypes = [_dp, _dp, C.c_int, C.c_double, C.c_int, _dp, _dp, C.c_double, C.c_double, _dp, _ip, _ip]
    L.orc_control_continuous_step.restype = C.c_int
    L.orc_control_continuous_step.argtypes = [_dp, _dp, _dp, C.c_int, C.c_double, C.c_double, C.c_int, _dp, _dp,
                                              C.c_double, C.c_double, _dp, _ip]
    L.orc_interval_limit.argtypes = [_dp, C.c_int, C.c_double, _dp]
    L.orc_solve_batch.argtypes = [_dp, _dp, C.c_long] + [_dp] * 6 + [_u8p, C.c_int, _dp, _dp, _dp, _dp, _dp, _u8p, _u8p, _u8p, C.c_int]
    L.orc_control_discrete_batch.argtypes = [_dp, _dp, C.c_long, _dp, _u8p, C.c_int, C.c_double, C.c_int, _dp, _dp,
                                             C.c_double, _dp, _u8p, _u8p, _u8p, C.c_int]
    L.orc_control_continuous_run_batch.argtypes = [_dp, _dp, C.c_long, C.c_long, _dp, C.c_int, C.c_double, C.c_double, C.c_int,
                                                   C.c_double, C.c_double, _dp, _u8p, _u8p, C.c_int, _dp]
    return L


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = _bind(C.CDLL(_LIB_PATH))
    return _lib


def native_lib():
    """The same C restatement compiled for THIS host's CPU (gcc -O3 -march=native, FP contraction still off, so the
    results are the portable build's bit for bit): the faster of the two CPU baselines bench.py reports.  The file name
    carries a hash of the CPU's model and flags, so a library built on another machine is never loaded."""
    global _native
    if _native is None:
        import hashlib

        try:
            with open("/proc/cpuinfo") as fh:
                info = [ln for ln in fh.read().splitlines() if ln.startswith(("model name", "flags"))][:2]
        except OSError:
            info = []
        tag = hashlib.sha256("\n".join(info).encode()).hexdigest()[:12]
        out_dir = os.path.join(_HERE, "_native")
        path = os.path.join(out_dir, f"librsik_oracle_{tag}.so")
        src = os.path.join(_HERE, "rsik_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            os.makedirs(out_dir, exist_ok=True)
            subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=c99", "-ffp-contract=off", "-fno-fast-math",
                                   "-fopenmp", src, "-o", path + ".tmp", "-shared", "-lm"])
            os.replace(path + ".tmp", path)
        _native = _bind(C.CDLL(path))
    return _native


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _u8(a):
    return None if a is None else a.ctypes.data_as(_u8p)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


class Arm:
    """orc_arm_t as a flat float64 array."""

    def __init__(self, arm="r_arm", singularity_offset=0.03, ik_parameters=None, elbow_limit=127.0, wrist_limit=42.5,
                 projection_margin=1e-8, backward_limit=0.02, normal_vector_margin=1e-7, singularity_limit_coeff=1.0):
        L = lib()
        self.name = arm
        self.is_left = int(arm == "l_arm")
        self.buf = np.zeros(L.orc_arm_ndoubles(), dtype=np.float64)
        if ik_parameters is None:
            assert (elbow_limit, wrist_limit, projection_margin, backward_limit, normal_vector_margin,
                    singularity_limit_coeff) == (127.0, 42.5, 1e-8, 0.02, 1e-7, 1.0)
            L.orc_arm_init_default(_d(self.buf), self.is_left, singularity_offset)
        else:
            p = arm[0]
            sp = _f64(ik_parameters[f"{p}_shoulder_position"])
            so = _f64(ik_parameters[f"{p}_shoulder_orientation"])
            tip = _f64(ik_parameters[f"{p}_tip_position"])
            L.orc_arm_init(_d(self.buf), self.is_left, _d(sp), _d(so), float(ik_parameters[f"{p}_upper_arm_size"]),
                           float(ik_parameters[f"{p}_forearm_size"]), _d(tip), float(elbow_limit), float(wrist_limit),
                           float(projection_margin), float(backward_limit), float(normal_vector_margin),
                           float(singularity_offset), float(singularity_limit_coeff))

    # field views (offsets follow orc_arm_t)
    def field(self, name):
        off = {"side": (0, 1), "shoulder_position": (1, 3), "shoulder_orientation_offset": (4, 3), "upper_arm_size": (7, 1),
               "forearm_size": (8, 1), "tip_position": (9, 3), "gripper_size": (12, 1), "max_arm_length": (13, 1),
               "projection_margin": (14, 1), "normal_vector_margin": (15, 1), "backward_limit": (16, 1),
               "elbow_limit": (17, 1), "shoulder_wrist_min_distance": (18, 1), "wrist_limit": (19, 1),
               "singularity_offset": (20, 1), "singularity_limit_coeff": (21, 1), "elbow_singularity_position": (22, 3),
               "wrist_singularity_position": (25, 3)}[name]
        v = self.buf[off[0]:off[0] + off[1]]
        return float(v[0]) if off[1] == 1 else v.copy()


class Solver:
    """One SymbolicIK instance worth of mutable state (orc_solver_t)."""

    def __init__(self, arm: Arm):
        self.arm = arm
        self.buf = np.zeros(lib().orc_solver_ndoubles(), dtype=np.float64)

    def is_reachable(self, pos, eul):
        pos, eul = _f64(pos), _f64(eul)
        ok = C.c_int(0)
        itv = np.zeros(2)
        st = lib().orc_is_reachable(_d(self.arm.buf), _d(self.buf), _d(pos), _d(eul), C.byref(ok), _d(itv))
        return bool(ok.value), itv, st

    def is_reachable_no_limits(self, pos, eul):
        pos, eul = _f64(pos), _f64(eul)
        return bool(lib().orc_is_reachable_no_limits(_d(self.arm.buf), _d(self.buf), _d(pos), _d(eul)))

    def get_elbow_position(self, theta):
        out = np.zeros(3)
        lib().orc_get_elbow_position(_d(self.buf), float(theta), _d(out))
        return out

    def get_joints(self, theta, previous_joints=None):
        j, e = np.zeros(7), np.zeros(3)
        pj = _f64(previous_joints)
        proj = lib().orc_get_joints(_d(self.arm.buf), _d(self.buf), float(theta), _d(pj), _d(j), _d(e))
        return j, e, bool(proj)

    def best_discrete_theta(self, interval, nb_search_points, preferred_theta, previous_theta=0.0):
        """utils.get_best_discrete_theta (utils.py:334-396) on the circle of the last is_reachable: (found, theta)."""
        itv, th = _f64(interval), np.zeros(1)
        ok = lib().orc_get_best_discrete_theta(_d(self.arm.buf), _d(self.buf), float(previous_theta), _d(itv),
                                               int(nb_search_points), float(preferred_theta), _d(th))
        return bool(ok), float(th[0])

    def best_theta_to_current_joints(self, current_joints, preferred_theta):
        cj = _f64(current_joints).ravel()
        return lib().orc_get_best_theta_to_current_joints(_d(self.arm.buf), _d(self.buf), _d(cj), int(cj.size),
                                                          float(preferred_theta))


def solve_batch(arm_r, arm_l, pos, eul, arm_id=None, theta_policy=0, theta_in=None, previous_joints=None, nthreads=1, L=None):
    pos, eul = _f64(pos), _f64(eul)
    n = pos.shape[0]
    cols = [np.ascontiguousarray(pos[:, k]) for k in range(3)] + [np.ascontiguousarray(eul[:, k]) for k in range(3)]
    joints = np.empty((n, 7)); interval = np.empty((n, 2)); elbow = np.empty((n, 3))
    reach = np.empty(n, dtype=np.uint8); state = np.empty(n, dtype=np.uint8); proj = np.empty(n, dtype=np.uint8)
    aid = None if arm_id is None else np.ascontiguousarray(arm_id, dtype=np.uint8)
    th = _f64(theta_in)
    pj = _f64(previous_joints)
    (L or lib()).orc_solve_batch(_d(arm_r.buf), _d(arm_l.buf), n, *[_d(c) for c in cols], _u8(aid), int(theta_policy), _d(th),
                          _d(pj), _d(joints), _d(interval), _d(elbow), _u8(reach), _u8(state), _u8(proj), int(nthreads))
    return dict(joints=joints, interval=interval, elbow=elbow, reachable=reach, state=state, projected=proj)


def control_discrete_batch(arm_r, arm_l, M, arm_id=None, nb_search_points=20, preferred_theta=-4 * np.pi / 6,
                           constrained_mode=0, previous_sol=None, current_joints=None,
                           orbita3d_max_angle=float(np.deg2rad(42.5)), nthreads=1, L=None):
    M = np.ascontiguousarray(M, dtype=np.float64).reshape(-1, 16)
    n = M.shape[0]
    if previous_sol is None:
        previous_sol = np.array([[0.0, 0.2617993877991494, -0.17453292519943295, 0.0, 0.0, 0.0, 0.0],
                                 [0.0, -0.2617993877991494, 0.17453292519943295, 0.0, 0.0, 0.0, 0.0]])
    ps = _f64(previous_sol)
    cj = _f64(current_joints)
    aid = None if arm_id is None else np.ascontiguousarray(arm_id, dtype=np.uint8)
    joints = np.empty((n, 7)); reach = np.empty(n, dtype=np.uint8); state = np.empty(n, dtype=np.uint8)
    em = np.empty(n, dtype=np.uint8)
    (L or lib()).orc_control_discrete_batch(_d(arm_r.buf), _d(arm_l.buf), n, _d(M), _u8(aid), int(nb_search_points),
                                     float(preferred_theta), int(constrained_mode), _d(ps), _d(cj),
                                     float(orbita3d_max_angle), _d(joints), _u8(reach), _u8(state), _u8(em), int(nthreads))
    return dict(joints=joints, reachable=reach, state=state, emergency=em)


class ContinuousState:
    def __init__(self, previous_theta, previous_sol):
        self.buf = np.zeros(11)
        self.buf[0] = previous_theta
        self.buf[1:8] = previous_sol
        self.buf[8] = 1.0   # init
        self.buf[9] = 0.0   # emergency_stop
        self.buf[10] = 1.0  # has_previous_sol

    @property
    def previous_theta(self):
        return float(self.buf[0])

    @property
    def previous_sol(self):
        return self.buf[1:8].copy()

    @property
    def emergency_stop(self):
        return bool(self.buf[9])


def interval_limit(arm, constrained_mode, preferred_theta=-4 * np.pi / 6):
    """control_ik.py:225-252: (interval_limit[0], interval_limit[1], preferred_theta) as the arm's side sees them."""
    out = np.zeros(3)
    lib().orc_interval_limit(_d(arm.buf), int(constrained_mode), float(preferred_theta), _d(out))
    return out


def control_continuous_step(arm, cs, M, timed_out, preferred_theta_arg, preferred_theta_self, constrained_mode,
                            current_joints, current_pose, d_theta_max=0.01,
                            orbita3d_max_angle=float(np.deg2rad(42.5))):
    M = _f64(M).reshape(16)
    cp = _f64(current_pose).reshape(16)
    cj = _f64(current_joints)
    j = np.zeros(7)
    ok = C.c_int(0)
    st = lib().orc_control_continuous_step(_d(arm.buf), _d(cs.buf), _d(M), int(timed_out), float(preferred_theta_arg),
                                           float(preferred_theta_self), int(constrained_mode), _d(cj), _d(cp),
                                           float(d_theta_max), float(orbita3d_max_angle), _d(j), C.byref(ok))
    return j, bool(ok.value), st


def control_continuous_run_batch(arm, states, M, first_step_timed_out=True, preferred_theta_arg=-4 * np.pi / 6,
                                 preferred_theta_self=-4 * np.pi / 6, constrained_mode=0, d_theta_max=0.01,
                                 orbita3d_max_angle=float(np.deg2rad(42.5)), nthreads=1, L=None, current_pose=None):
    """M [n_steps, n_traj, 4, 4]; states [n_traj, 11] float64 (orc_cont_state_t rows: previous_theta, previous_sol[7], init,
    emergency_stop, has_previous_sol), updated in place.  current_pose: the 4x4 pose every trajectory's first step is called with
    (what a fresh ControlIK holds in previous_pose), or None for the trajectory's own first matrix."""
    M = np.ascontiguousarray(M, dtype=np.float64)
    n_steps, n_traj = M.shape[:2]
    assert states.shape == (n_traj, 11) and states.dtype == np.float64 and states.flags.c_contiguous
    joints = np.empty((n_steps, n_traj, 7)); reach = np.empty((n_steps, n_traj), dtype=np.uint8)
    state = np.empty((n_steps, n_traj), dtype=np.uint8)
    (L or lib()).orc_control_continuous_run_batch(_d(arm.buf), _d(states), n_traj, n_steps, _d(M.reshape(-1)), int(bool(first_step_timed_out)),
                                                  float(preferred_theta_arg), float(preferred_theta_self), int(constrained_mode),
                                                  float(d_theta_max), float(orbita3d_max_angle), _d(joints), _u8(reach), _u8(state),
                                                  int(nthreads), _d(None if current_pose is None else _f64(current_pose).reshape(16)))
    return dict(joints=joints, reachable=reach, state=state)


def euler_from_matrix_xyz(M):
    """utils.get_euler_from_homogeneous_matrix's angles for a batch of [n,4,4] / [n,3,3] matrices."""
    M = np.asarray(M, dtype=np.float64)
    R = np.ascontiguousarray(M[:, :3, :3]).reshape(-1, 9)
    out = np.empty((R.shape[0], 3))
    for k in range(R.shape[0]):
        lib().orc_euler_from_matrix_xyz(_d(np.ascontiguousarray(R[k])), _d(out[k]))
    return out
