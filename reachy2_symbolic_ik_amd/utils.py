"""The reference's utils module for callers that import its helpers (reachy2_symbolic_ik/utils.py; src/example/test_ik.py:16-21,
test_go_to.py:10-13): same names, arguments, return values and messages.  Every number is computed by the device — each function
is one rsik_stage launch (include/rsik.h, RSIK_STAGE_*: the reference's own sequence of operations on the arguments given) or
rsik_matrix_to_pose — there is no CPU implementation of the path in this package; only the packing of arguments, the reference's
message texts and the URDF reader (constants.py) are host code.

The functions share one device context, created on first use on the current device (`set_default_solver` hands over another:
an existing HipSolver, or a device index); calls are serialised by a lock, 20-30 us each (a launch and a stream synchronisation
through pinned host rows: these are scalar utilities — batches go through `HipSolver.stage`).  Importing this module needs no GPU;
calling a helper does.  Not provided: the matplotlib drawing helpers (show_*), dead code (get_best_continuous_theta,
get_best_continuous_theta2 and tend_to_preferred_theta live inside the continuous control kernels and take closures that only
exist there) — SURVEY section 2's out-of-scope lines.
"""
from __future__ import annotations

import copy
import threading
from typing import Any, List, Optional, Tuple

import numpy as np
import numpy.typing as npt

from . import _abi
from .constants import get_ik_parameters_from_urdf, parse_vector  # noqa: F401  (utils.py:661-694)

_lock = threading.Lock()
_default: dict = {}


def set_default_solver(solver_or_device: Any = None) -> None:
    """The device context the module's helpers run on: a HipSolver, a device index / torch.device, or None (= created on first use
    on the current device)."""
    with _lock:
        _default.clear()
        if solver_or_device is not None:
            _default["given"] = solver_or_device


def _io():
    """(solver, pinned input row, pinned output row, their numpy views): created once, under the lock."""
    if "solver" not in _default:
        import torch

        from .backend import HipSolver

        given = _default.get("given")
        _default["solver"] = given if isinstance(given, HipSolver) else HipSolver(given)
        _default["in"] = torch.empty((1, _abi.STAGE_IN_MAX), dtype=torch.float64).pin_memory()
        _default["out"] = torch.empty((1, _abi.STAGE_OUT_MAX), dtype=torch.float64).pin_memory()
        _default["in_np"], _default["out_np"] = _default["in"].numpy(), _default["out"].numpy()
    return _default


def _stage(op: int, *operands: Any) -> np.ndarray:
    """One row through rsik_stage (pinned host memory the device addresses directly: no upload, no download)."""
    import torch

    need_in, need_out = _abi.STAGE_ROW[op]
    flat = np.concatenate([np.asarray(a, dtype=np.float64).reshape(-1) for a in operands])
    if flat.size != need_in:
        raise ValueError(f"stage {op} takes {need_in} numbers, got {flat.size}")
    with _lock:
        io = _io()
        sv = io["solver"]
        io["in_np"][0, :need_in] = flat
        with torch.cuda.device(sv.device):
            sv._bind_stream()
            sv._check(sv.lib.rsik_stage(sv._h, int(op), 1, 0, io["in"].data_ptr(), _abi.STAGE_IN_MAX, io["out"].data_ptr(), _abi.STAGE_OUT_MAX))
            torch.cuda.current_stream(sv.device).synchronize()
        return io["out_np"][0, :need_out].copy()


def make_homogenous_matrix_from_rotation_matrix(
    position: npt.NDArray[np.float64], rotation_matrix: npt.NDArray[np.float64]
) -> npt.NDArray[np.float64]:
    """4x4 homogeneous matrix from a 3x3 rotation and a position (utils.py:12-23): packing, no arithmetic."""
    M = np.eye(4)
    M[:3, :3] = np.asarray(rotation_matrix, dtype=np.float64)[:3, :3]
    M[:3, 3] = np.asarray(position, dtype=np.float64)[:3]
    return M


def rotation_matrix_from_vector(vect: npt.NDArray[np.float64]) -> npt.NDArray[np.float64]:
    """The rotation that takes e_x to vect / |vect| (utils.py:59-81)."""
    return _stage(_abi.STAGE_ROTATION_FROM_VECTOR, vect).reshape(3, 3)


def get_euler_from_homogeneous_matrix(
    homogeneous_matrix: npt.NDArray[np.float64], degrees: bool = False
) -> Tuple[npt.NDArray[np.float64], npt.NDArray[np.float64]]:
    """(position, extrinsic xyz Euler angles) of a 4x4 pose matrix (utils.py:84-90): rsik_matrix_to_pose — SciPy's nearest-rotation,
    matrix -> quaternion -> Euler algorithms with its gimbal-lock rule, pinned by the reference's own angles (G8)."""
    import torch

    M = np.asarray(homogeneous_matrix, dtype=np.float64)
    with _lock:
        sv = _io()["solver"]
        m12 = torch.as_tensor(np.concatenate([M[:3, :3].reshape(9), M[:3, 3]]).reshape(12, 1)).to(sv.device)
        pose = sv.matrix_to_pose(m12, identity_shortcut=False)
        torch.cuda.current_stream(sv.device).synchronize()
        eul = pose[3:6, 0].cpu().numpy()
    return M[:3, 3], (np.degrees(eul) if degrees else eul)


def angle_diff(a: float, b: float) -> float:
    """The smallest signed distance between two angles (utils.py:486-490)."""
    return float(_stage(_abi.STAGE_ANGLE_DIFF, a, b)[0])


def is_valid_angle(angle: float, interval: npt.NDArray[np.float64]) -> bool:
    """utils.py:468-474."""
    return bool(_stage(_abi.STAGE_IS_VALID_ANGLE, angle, interval[0], interval[1])[0] != 0.0)


def limit_theta_to_interval(theta: float, previous_theta: float, interval: npt.NDArray[np.float64]) -> Tuple[float, str]:
    """utils.py:93-112: theta wrapped to (-pi, pi], or the nearer end of the interval."""
    o = _stage(_abi.STAGE_LIMIT_THETA_TO_INTERVAL, theta, previous_theta, interval[0], interval[1])
    if o[1] != 0.0:
        return float(o[0]), "theta in interval"
    # (the reference hands back interval[k] itself, whatever its type)
    return (interval[1] if o[0] == float(interval[1]) else interval[0]), "theta not in interval"


def is_elbow_ok(
    elbow_position: npt.NDArray[np.float64],
    side: int,
    singularity_offset: float,
    singularity_limit_coeff: float,
    elbow_singularity_position: npt.NDArray[np.float64],
) -> bool:
    """utils.py:443-465."""
    return bool(_stage(_abi.STAGE_IS_ELBOW_OK, np.asarray(elbow_position, dtype=np.float64)[:3], side, singularity_offset,
                       singularity_limit_coeff, np.asarray(elbow_singularity_position, dtype=np.float64)[:3])[0] != 0.0)


def allow_multiturn(new_joints: List[float], prev_joints: List[float], name: str) -> List[float]:
    """utils.py:493-505: previous + angle_diff(new, previous), joint by joint."""
    n = len(new_joints)
    if n > 7:
        raise ValueError("allow_multiturn: at most 7 joints")
    a, b = np.zeros(7), np.zeros(7)
    a[:n], b[:n] = np.asarray(new_joints, dtype=np.float64), np.asarray(prev_joints, dtype=np.float64)[:n]
    out = copy.deepcopy(new_joints)
    res = _stage(_abi.STAGE_ALLOW_MULTITURN, a, b)
    for i in range(n):
        out[i] = float(res[i])
    return out


def limit_orbita3d_joints(joints: List[float], orbita3D_max_angle: float) -> List[float]:
    """utils.py:508-519: the three angles cast into the Orbita3D cone (intrinsic XYZ -> ZYZ, clamp, back; SciPy's gimbal rule)."""
    o = _stage(_abi.STAGE_LIMIT_ORBITA3D_JOINTS, joints[0], joints[1], joints[2], orbita3D_max_angle)
    return [float(o[0]), float(o[1]), float(o[2])]


def limit_orbita3d_joints_wrist(joints: List[float], orbita3D_max_angle: float) -> List[float]:
    """utils.py:522-532."""
    joints = copy.deepcopy(joints)
    joints[4:7] = limit_orbita3d_joints(joints[4:7], orbita3D_max_angle)
    return joints


_LIMIT_TEXT = ((_abi.EMERGENCY_SHOULDER_PITCH, 0, "EMERGENCY STOP: shoulder pitch limit reached"),
               (_abi.EMERGENCY_ELBOW_YAW, 2, "EMERGENCY STOP: elbow yaw limit reached"),
               (_abi.EMERGENCY_WRIST_YAW, 6, "EMERGENCY STOP: wrist yaw limit reached"))


def multiturn_safety_check(
    joints: List[float], shoulder_pitch_limit: float, elbow_yaw_limit: float, wrist_yaw_limit: float, emergency_state: str
) -> Tuple[List[float], bool, str]:
    """utils.py:535-568: joints 0, 2 and 6 clamped to their limits; the reference's messages appended."""
    o = _stage(_abi.STAGE_MULTITURN_SAFETY_CHECK, np.asarray(joints, dtype=np.float64)[:7], shoulder_pitch_limit, elbow_yaw_limit, wrist_yaw_limit)
    out = copy.deepcopy(joints)
    cause = int(o[7])
    for bit, k, text in _LIMIT_TEXT:
        if cause & bit:
            out[k] = float(o[k])
            emergency_state += "\n" + text
    return out, cause != 0, emergency_state


def continuity_check(
    joints: npt.NDArray[np.float64], previous_joints: npt.NDArray[np.float64], max_angulare_change: List[float], emergency_state: str
) -> Tuple[npt.NDArray[np.float64], bool, str]:
    """utils.py:571-589."""
    o = _stage(_abi.STAGE_CONTINUITY_CHECK, np.asarray(joints, dtype=np.float64)[:7], np.asarray(previous_joints, dtype=np.float64)[:7],
               np.asarray(max_angulare_change, dtype=np.float64)[:7])
    if o[7] != 0.0:
        emergency_state += f"\n EMERGENCY STOP: joints are not continuous \n previous_joints: {previous_joints} \n joints: {joints}"
        return np.array(previous_joints), True, emergency_state
    return joints, False, emergency_state


def get_best_discrete_theta(
    previous_theta: float,
    interval: npt.NDArray[np.float64],
    get_elbow_position: Any,
    nb_search_points: int,
    preferred_theta: float,
    arm: str,
    singularity_offset: float,
    singularity_limit_coeff: float,
    elbow_singularity_position: npt.NDArray[np.float64],
) -> Tuple[bool, float, str]:
    """utils.py:334-396: the valid theta of the interval closest to preferred_theta.  `get_elbow_position` is the bound method of
    a SymbolicIK solver, as in the reference's own call (control_ik.py:424-434): the search runs on the device over the
    intersection circle that solver holds (its `intersection_circle`, which the reference's method reads too, symbolic_ik.py:684-695).
    The returned text carries what the reference's does up to the grid ("debug_dict" is the reference's debugging aid: not rebuilt)."""
    owner = getattr(get_elbow_position, "__self__", None)
    circle = getattr(owner, "intersection_circle", None)
    if circle is None:
        raise TypeError("get_best_discrete_theta: get_elbow_position must be the get_elbow_position method of a SymbolicIK solver "
                        "(the search reads the solver's intersection_circle)")
    side = -1 if arm == "l_arm" else 1
    o = _stage(_abi.STAGE_BEST_DISCRETE_THETA, previous_theta, interval[0], interval[1], nb_search_points, preferred_theta, side,
               singularity_offset, singularity_limit_coeff, np.asarray(elbow_singularity_position, dtype=np.float64)[:3],
               np.asarray(circle[0], dtype=np.float64)[:3], circle[1], np.asarray(circle[2], dtype=np.float64)[:3])
    state = f"{arm}" + "\n" + f"interval: {interval}, preferred_theta: {preferred_theta}"
    if o[2] != 0.0:
        return True, preferred_theta, state + "\n" + "preferred_theta worked!"
    if o[0] != 0.0:
        return True, float(o[1]), state
    return False, previous_theta, state


def utils_on_device() -> Optional[Any]:
    """The HipSolver the helpers run on (None before the first call)."""
    return _default.get("solver")
