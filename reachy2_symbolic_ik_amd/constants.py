"""Host-side per-arm constants (what SymbolicIK.__init__ computes once) packed for the HIP kernels.

Reference: symbolic_ik.py:26-83 (constructor), utils.py:26-43 (get_singularity_position),
symbolic_ik.py:728-738 (shoulder frame), symbolic_ik.py:653-672 (pose-independent half of
make_elbow_projection), utils.py:661-694 (URDF -> ik_parameters).

Layout of the packed block = the RSIK_C_* offsets of include/rsik.h.
"""
from __future__ import annotations

import math
import xml.etree.ElementTree as ET
from io import StringIO
from typing import Any, Dict, List

import numpy as np

# include/rsik.h offsets
C_SHOULDER, C_UPPER_ARM, C_FOREARM, C_TIPL, C_MAX_LEN, C_MIN_DIST, C_BACKWARD = 0, 3, 4, 5, 8, 9, 10
C_PROJ_MARGIN, C_NORMAL_MARGIN, C_UPF, C_WRIST_R, C_WRIST_AX, C_MST, C_TSH, C_ES = 11, 12, 13, 14, 15, 16, 25, 28
C_SING_OFFSET, C_SING_COEFF, C_ELBOW_LIMIT, C_SIDE, C_PLANE_P, C_PLANE_N = 31, 32, 33, 34, 35, 38
C_PROJ_CENTER, C_PROJ_RADIUS, C_TIP_Z = 41, 44, 45
C_INV_U, C_INV_F, C_INV_TIPZ, C_INV_GRIP, C_MAX_LEN_SQ, C_INV_MIN_DIST, C_PLANE_K, ARM_CONSTS_COUNT = 46, 47, 48, 49, 50, 51, 52, 53

ARM_IDS = {"r_arm": 0, "l_arm": 1}

STATE_STRINGS = (
    "reachable",
    "Pose out of reach",
    "Backward pose",
    "wrist out of range",
    "limited by wrist",
    "out of reach - should not happen",
    "limited by shoulder",
    "",
    "emergency stop",
    "not reachable without limits",  # RSIK_STATE_NOT_REACHABLE_NO_LIMITS: where the reference raises (control_ik.py:385-387)
    "invalid input",  # RSIK_STATE_INVALID_INPUT: the goal holds a NaN / an infinity — where the reference raises LinAlgError (symbolic_ik.py:580)
)


def default_ik_parameters() -> Dict[str, Any]:
    """symbolic_ik.py:40-51."""
    return {
        "r_shoulder_position": np.array([0.0, -0.2, 0.0]),
        "r_shoulder_orientation": [-15, 0, 10],
        "r_upper_arm_size": np.float64(0.28),
        "r_forearm_size": np.float64(0.28),
        "r_tip_position": np.array([-0.0, 0.0, 0.10]),
        "l_shoulder_position": np.array([0.0, 0.2, 0.0]),
        "l_shoulder_orientation": [15, 0, -10],
        "l_upper_arm_size": np.float64(0.28),
        "l_forearm_size": np.float64(0.28),
        "l_tip_position": np.array([-0.0, 0.0, 0.10]),
    }


def _rx(a: float) -> np.ndarray:
    c, s = np.cos(a), np.sin(a)
    return np.array([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]])


def _ry(a: float) -> np.ndarray:
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])


def _rz(a: float) -> np.ndarray:
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def euler_xyz_extrinsic(e) -> np.ndarray:
    """Rotation matrix of scipy's from_euler("xyz", e): Rz(e2) Ry(e1) Rx(e0)."""
    return _rz(float(e[2])) @ _ry(float(e[1])) @ _rx(float(e[0]))


class ArmGeometry:
    """The numbers SymbolicIK exposes as attributes, plus the packed kernel block."""

    def __init__(
        self,
        arm: str,
        ik_parameters: Dict[str, Any],
        elbow_limit: float = 127,
        wrist_limit: float = 42.5,
        projection_margin: float = 1e-8,
        backward_limit: float = 0.02,
        normal_vector_margin: float = 1e-7,
        singularity_offset: float = 0.03,
        singularity_limit_coeff: float = 1.0,
    ) -> None:
        if arm not in ("r_arm", "l_arm"):
            raise ValueError("arm should be either 'r_arm' or 'l_arm'")
        p = arm[0]
        self.arm = arm
        self.side = 1.0 if arm == "r_arm" else -1.0
        self.shoulder_position = np.array(ik_parameters[f"{p}_shoulder_position"], dtype=np.float64)
        self.shoulder_orientation_offset = ik_parameters[f"{p}_shoulder_orientation"]
        self.upper_arm_size = np.float64(ik_parameters[f"{p}_upper_arm_size"])
        self.forearm_size = np.float64(ik_parameters[f"{p}_forearm_size"])
        self.tip_position = np.array(ik_parameters[f"{p}_tip_position"], dtype=np.float64)
        self.gripper_size = np.linalg.norm(self.tip_position)
        self.max_arm_length = self.upper_arm_size + self.forearm_size + self.gripper_size
        self.projection_margin = projection_margin
        self.normal_vector_margin = normal_vector_margin
        self.backward_limit = backward_limit
        self.elbow_limit = elbow_limit
        u, f = self.upper_arm_size, self.forearm_size
        self.shoulder_wrist_min_distance = np.sqrt(u**2 + f**2 - 2 * u * f * np.cos(np.radians(180 - elbow_limit)))
        self.wrist_limit = wrist_limit
        self.singularity_offset = singularity_offset
        self.singularity_limit_coeff = singularity_limit_coeff
        # utils.py:26-43
        off_rad = np.radians(np.array(self.shoulder_orientation_offset, dtype=np.float64))
        R_off = euler_xyz_extrinsic(off_rad)
        self.elbow_singularity_position = R_off @ np.array([0.0, -u * self.side, 0.0]) + self.shoulder_position
        self.wrist_singularity_position = R_off @ np.array([0.0, -(u + f) * self.side, 0.0]) + self.shoulder_position
        self._R_off = R_off

    def pack(self) -> np.ndarray:
        c = np.zeros(ARM_CONSTS_COUNT, dtype=np.float64)
        s, u, f = self.shoulder_position, float(self.upper_arm_size), float(self.forearm_size)
        c[C_SHOULDER:C_SHOULDER + 3] = s
        c[C_UPPER_ARM], c[C_FOREARM] = u, f
        c[C_TIPL:C_TIPL + 3] = [-self.tip_position[0], self.tip_position[1], self.tip_position[2]]
        c[C_MAX_LEN] = self.max_arm_length
        c[C_MIN_DIST] = self.shoulder_wrist_min_distance
        c[C_BACKWARD] = self.backward_limit
        c[C_PROJ_MARGIN] = self.projection_margin
        c[C_NORMAL_MARGIN] = self.normal_vector_margin
        c[C_UPF] = self.upper_arm_size + self.forearm_size
        rw = np.sin(np.radians(self.wrist_limit)) * self.forearm_size  # symbolic_ik.py:413
        c[C_WRIST_R] = rw
        c[C_WRIST_AX] = np.sqrt(self.forearm_size**2 - rw**2)  # symbolic_ik.py:414
        # symbolic_ik.py:728-738
        M_torso_shoulder = self._R_off @ euler_xyz_extrinsic([0.0, np.pi / 2, 0.0])
        MsT = M_torso_shoulder.T
        c[C_MST:C_MST + 9] = MsT.reshape(9)
        c[C_TSH:C_TSH + 3] = (-MsT) @ s
        c[C_ES:C_ES + 3] = self.elbow_singularity_position
        c[C_SING_OFFSET] = self.singularity_offset
        c[C_SING_COEFF] = self.singularity_limit_coeff
        c[C_ELBOW_LIMIT] = np.radians(self.elbow_limit)
        c[C_SIDE] = self.side
        # symbolic_ik.py:653-672 (pose independent)
        alpha = np.arctan2(-self.singularity_limit_coeff, 1)
        Ml = euler_xyz_extrinsic([0.0, alpha, 0.0])
        Pl = Ml @ np.array([0.0, 0.0, -self.singularity_offset]) + self.elbow_singularity_position
        n1 = Ml @ np.array([1.0, 0.0, 0.0]) + Pl
        n2 = Ml @ np.array([0.0, 1.0, 0.0]) + Pl
        v3 = np.cross(n1 - Pl, n2 - Pl)
        v3 = v3 / np.linalg.norm(v3)
        pc = s - np.dot(s - Pl, v3) * v3
        with np.errstate(invalid="ignore"):
            pr = np.sqrt(self.upper_arm_size**2 - np.linalg.norm(s - pc) ** 2)
        c[C_PLANE_P:C_PLANE_P + 3] = Pl
        c[C_PLANE_N:C_PLANE_N + 3] = v3
        c[C_PROJ_CENTER:C_PROJ_CENTER + 3] = pc
        c[C_PROJ_RADIUS] = pr
        c[C_TIP_Z] = self.tip_position[2]
        with np.errstate(divide="ignore"):
            c[C_INV_U] = 1.0 / self.upper_arm_size
            c[C_INV_F] = 1.0 / self.forearm_size
            c[C_INV_TIPZ] = 1.0 / abs(self.tip_position[2])
            c[C_INV_GRIP] = 1.0 / self.gripper_size
            c[C_INV_MIN_DIST] = 1.0 / self.shoulder_wrist_min_distance
        c[C_MAX_LEN_SQ] = sqrt_threshold(self.max_arm_length)
        es = self.elbow_singularity_position
        c[C_PLANE_K] = es[2] - self.singularity_offset - self.singularity_limit_coeff * es[0]
        return c


def sqrt_threshold(limit: float) -> float:
    """Largest double x with sqrt(x) <= limit (sqrt correctly rounded, hence monotonic): the kernels test
    `v.v > x` instead of `sqrt(v.v) > limit` (symbolic_ik.py:290) with the same outcome for every input."""
    limit = float(limit)
    if not (limit > 0.0) or not math.isfinite(limit):
        return limit * limit
    x = limit * limit
    while math.sqrt(x) > limit:
        x = math.nextafter(x, 0.0)
    while math.sqrt(math.nextafter(x, math.inf)) <= limit:
        x = math.nextafter(x, math.inf)
    return x


def parse_vector(vector_str: str) -> np.ndarray:
    """utils.py:693-694."""
    return np.array(list(map(float, vector_str.split())))


def get_ik_parameters_from_urdf(urdf_str: str, arm: List[str]) -> Dict[str, Any]:
    """utils.py:661-690: the four fixed-joint origins per arm that define the IK geometry."""
    root = ET.parse(StringIO(urdf_str)).getroot()
    out: Dict[str, Any] = {}
    for joint in root.findall("joint"):
        jname = joint.attrib["name"]
        for name in arm:
            if jname == f"{name}_shoulder_base_joint":
                origin = joint.find("origin").attrib  # type: ignore[union-attr]
                out[f"{name}_shoulder_position"] = parse_vector(origin["xyz"])
                orientation = parse_vector(origin["rpy"])
                orientation[0] += -np.pi / 2 if name == "r" else np.pi / 2
                out[f"{name}_shoulder_orientation"] = np.degrees(orientation)
            elif jname == f"{name}_elbow_base_joint":
                position = parse_vector(joint.find("origin").attrib["xyz"])  # type: ignore[union-attr]
                out[f"{name}_upper_arm_size"] = position[2]
                out[f"{name}_elbow_roll_offset"] = -position[0]
            elif jname == f"{name}_wrist_base_joint":
                position = parse_vector(joint.find("origin").attrib["xyz"])  # type: ignore[union-attr]
                out[f"{name}_forearm_size"] = position[2]
                out[f"{name}_wrist_pitch_offset"] = -position[1]
            elif jname == f"{name}_tip_joint":
                out[f"{name}_tip_position"] = parse_vector(joint.find("origin").attrib["xyz"])  # type: ignore[union-attr]
    return out


def interval_limit_for(arm: str, constrained_mode: str, preferred_theta: float):
    """control_ik.py:225-252: theta limit interval per constrained mode, mirrored for the left arm."""
    if constrained_mode == "unconstrained":
        lim = np.array([3 * np.pi / 4, -2 * np.pi / 6])
    elif constrained_mode == "low_elbow":
        lim = np.array([-4 * np.pi / 5, 0])
    else:
        raise UnboundLocalError("local variable 'interval_limit' referenced before assignment")
    if arm.startswith("l"):
        lim = np.array([-np.pi - lim[1], -np.pi - lim[0]])
        for k in (0, 1):
            if lim[k] < -np.pi:
                lim[k] = lim[k] % (2 * np.pi)
        for k in (0, 1):
            if lim[k] > np.pi:
                lim[k] = lim[k] % (-2 * np.pi)
        preferred_theta = -np.pi - preferred_theta
    return lim, preferred_theta
