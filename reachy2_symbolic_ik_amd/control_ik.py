"""ControlIK — drop-in for reachy2_symbolic_ik.control_ik.ControlIK (control_ik.py:27-497) on MI355X.

`symbolic_inverse_kinematics(name, M, "discrete", ...)` keeps the reference signature and return
tuple; `symbolic_inverse_kinematics_batch` is the MI355X-native form (matrices laid out SoA in HBM,
wave-cooperative theta sweep in the kernel).  No CPU fallback.
"""
from __future__ import annotations

import copy
import os
import time
from typing import Any, Dict, Optional, Tuple

import numpy as np
import numpy.typing as npt
import torch

from . import _abi
from .backend import HipSolver
from .constants import ARM_IDS, STATE_STRINGS, get_ik_parameters_from_urdf
from .symbolic_ik import SymbolicIK

DEFAULT_CURRENT_JOINTS = [
    [0.0, 0.2617993877991494, -0.17453292519943295, 0.0, 0.0, 0.0, 0.0],
    [0.0, -0.2617993877991494, 0.17453292519943295, 0.0, 0.0, 0.0, 0.0],
]
DEFAULT_CURRENT_POSE = [
    np.array([[1, 0, 0, 0], [0, 1, 0, -0.2], [0, 0, 1, -0.66], [0, 0, 0, 1]]),
    np.array([[1, 0, 0, 0], [0, 1, 0, 0.2], [0, 0, 1, -0.66], [0, 0, 0, 1]]),
]


def matrices_to_m12_soa(M: Any, device: torch.device) -> torch.Tensor:
    """[n,4,4] homogeneous matrices (or an already packed [12,n]) -> contiguous [12,n] float64 on `device`
    (rows R00..R22 row-major, then tx, ty, tz)."""
    t = M if isinstance(M, torch.Tensor) else torch.as_tensor(np.asarray(M, dtype=np.float64))
    t = t.to(device=device, dtype=torch.float64)
    if t.dim() == 2 and t.shape[0] == 12:
        return t.contiguous()
    if t.dim() == 2 and tuple(t.shape) == (4, 4):
        t = t.unsqueeze(0)
    if not (t.dim() == 3 and tuple(t.shape[1:]) == (4, 4)):
        raise ValueError("M must have shape [n,4,4], [4,4] or [12,n]")
    rot = t[:, :3, :3].reshape(t.shape[0], 9)
    tr = t[:, :3, 3]
    return torch.cat([rot, tr], dim=1).t().contiguous()


def emergency_messages(cause: int, previous_joints: Any = None, joints: Any = None) -> str:
    """What the reference appends to ControlIK.emergency_state for the RSIK_EMERGENCY_* cause bits of one call:
    utils.multiturn_safety_check (utils.py:544-566), then utils.continuity_check (utils.py:584-586), in that order."""
    text = ""
    for bit, joint in ((_abi.EMERGENCY_SHOULDER_PITCH, "shoulder pitch"), (_abi.EMERGENCY_ELBOW_YAW, "elbow yaw"),
                       (_abi.EMERGENCY_WRIST_YAW, "wrist yaw")):
        if cause & bit:
            text += "\n" + f"EMERGENCY STOP: {joint} limit reached"
    if cause & _abi.EMERGENCY_CONTINUITY:
        text += f"\n EMERGENCY STOP: joints are not continuous \n previous_joints: {np.asarray(previous_joints)} \n joints: {np.asarray(joints)}"
    return text


def angle_diff(a: float, b: float) -> float:
    """utils.py:486-490."""
    d = a - b
    return ((d + np.pi) % (2 * np.pi)) - np.pi


def get_best_theta_to_current_joints(get_joints: Any, nb_search_points: int, current_joints: Any, arm: str,
                                     preferred_theta: float) -> Tuple[float, str]:
    """utils.py:267-331: ternary search over the whole circle for the theta whose joints are closest to
    `current_joints`.  `current_joints` is used exactly as handed over — including ControlIK.__init__'s
    list-of-both-arms form (control_ik.py:152-158), which NumPy broadcasting turns into a 2x7 comparison (Q15)."""
    current_joints = copy.deepcopy(current_joints)
    low, high = (-np.pi, np.pi) if arm != "l_arm" else (0, 2 * np.pi)
    tolerance = 0.01

    def distance(joints: Any) -> float:
        return float(np.linalg.norm([angle_diff(joints[i], np.asarray(current_joints[i])) for i in range(len(current_joints))]))

    joints, _ = get_joints(preferred_theta)
    if distance(joints) < tolerance:
        return preferred_theta, "preferred_theta worked!"
    while (high - low) > tolerance:
        mid1 = low + (high - low) / 3
        mid2 = high - (high - low) / 3
        j1, _ = get_joints(mid1)
        j2, _ = get_joints(mid2)
        if distance(j1) < distance(j2):
            high = mid2
        else:
            low = mid1
    best_theta = (low + high) / 2
    get_joints(best_theta)  # utils.py:324 (the reference evaluates it once more; state-mutating, Q1)
    return best_theta, f" \n low = {low}, high = {high}"


class ControlIK:
    def __init__(
        self,
        current_joints: list = DEFAULT_CURRENT_JOINTS,
        current_pose: list = DEFAULT_CURRENT_POSE,
        logger: Any = None,
        urdf: str = "",
        urdf_path: str = "",
        reachy_model: str = "full_kit",
        is_dvt: bool = False,
        device: Any = None,
        solver: Optional[HipSolver] = None,
    ) -> None:
        # public attributes of the reference object (control_ik.py:60-84)
        self.symbolic_ik_solver: Dict[str, SymbolicIK] = {}
        self.preferred_theta: Dict[str, float] = {}
        self.previous_theta: Dict[str, float] = {}
        self.previous_sol: Dict[str, npt.NDArray[np.float64]] = {}
        self.previous_pose: Dict[str, npt.NDArray[np.float64]] = {}
        self.last_call_t: Dict[str, float] = {}
        self.logger = logger
        self.call_timeout, self.nb_search_points = 0.2, 20
        self.emergency_state, self.emergency_stop, self.init = "", False, True
        self.singularity_offset, self.singularity_limit_coeff = (0.03 if is_dvt else -1.01), 1.0
        self.orbita3D_max_angle = np.deg2rad(42.5)
        # not in the reference: how the kernels treat control_ik.py:215's matrix -> Euler -> matrix round trip:
        # "auto" makes it only where it changes the result (non-orthonormal or gimbal-lock matrices), "always" / "never"
        self.euler_roundtrip = "auto"
        if is_dvt:
            self._say("DVT mode activated", 0.1)

        arms = self._arm_prefixes(reachy_model, self._read_urdf(urdf, urdf_path))
        self._solver = solver if solver is not None else (HipSolver(device) if arms[0] else None)
        for prefix in arms[0]:
            self._add_arm(prefix, arms[1], current_joints, current_pose)

    _MODEL_ARMS = {"full_kit": ["r", "l"], "headless": ["r", "l"], "starter_kit_right": ["r"], "starter_kit_left": ["l"],
                   "mini": []}

    def _say(self, message: str, throttle: float) -> None:
        """rclpy-style logger if one was injected, else stdout (control_ik.py:72-75, 200-209)."""
        if self.logger is None:
            print(message)
        else:
            self.logger.info(message, throttle_duration_sec=throttle)

    @staticmethod
    def _read_urdf(urdf: str, urdf_path: str) -> str:
        """control_ik.py:86-97: URDF text, or the file at `urdf_path` relative to this package."""
        if not urdf and not urdf_path:
            raise ValueError("No URDF provided")
        if urdf:
            return urdf
        path = os.path.join(os.path.dirname(__file__), urdf_path)
        text = ""
        if os.path.isfile(path) and os.path.getsize(path) > 0:
            with open(path, "r") as fh:
                text = fh.read()
        if not text:
            raise ValueError("Empty URDF file")
        return text

    @classmethod
    def _arm_prefixes(cls, reachy_model: str, urdf: str):
        """control_ik.py:98-112: which arms the model has, and their IK parameters from the URDF."""
        if reachy_model not in cls._MODEL_ARMS:
            raise ValueError(f"Unknown Reachy model {reachy_model}")
        prefixes = cls._MODEL_ARMS[reachy_model]
        try:
            return prefixes, get_ik_parameters_from_urdf(urdf, prefixes)
        except Exception as e:
            raise ValueError(f"Error while parsing URDF: {e}")

    def _add_arm(self, prefix: str, ik_parameters: Dict[str, Any], current_joints: list, current_pose: list) -> None:
        """control_ik.py:114-160: one SymbolicIK per arm and the start-up theta that best matches current_joints."""
        arm, k = f"{prefix}_arm", "rl".index(prefix)
        kwargs = dict(ik_parameters=ik_parameters) if ik_parameters else dict(wrist_limit=np.rad2deg(self.orbita3D_max_angle))
        self.symbolic_ik_solver[arm] = SymbolicIK(arm=arm, singularity_offset=self.singularity_offset,
                                                  singularity_limit_coeff=self.singularity_limit_coeff,
                                                  solver=self._solver, **kwargs)
        base = -4 * np.pi / 6
        self.preferred_theta[arm] = base if k == 0 else -np.pi - base
        self.previous_sol[arm] = np.array(current_joints[k])
        self.previous_pose[arm] = current_pose[k]
        self.last_call_t[arm] = 0.0
        _, _, joints_of_theta = self.symbolic_ik_solver[arm].is_reachable_no_limits(self._matrix_to_pose(current_pose[k]))
        # the reference hands over BOTH arms' joint lists here (Q15); kept, since it fixes previous_theta's start value
        self.previous_theta[arm], _ = get_best_theta_to_current_joints(joints_of_theta, 20, current_joints, arm,
                                                                       self.preferred_theta[arm])

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _matrix_to_pose(M: npt.NDArray[np.float64]) -> npt.NDArray[np.float64]:
        """control_ik.py:212-217 / 142-147: identity shortcut, else extrinsic xyz Euler angles of M[:3,:3]."""
        M = np.asarray(M, dtype=np.float64)
        if np.allclose(M[:3, :3], np.eye(3)):
            return np.array([M[:3, 3], [0, 0, 0]], dtype=np.float64)
        from scipy.spatial.transform import Rotation

        return np.array([M[:3, 3], Rotation.from_matrix(M[:3, :3]).as_euler("xyz")])

    def _previous_sol_2x7(self) -> np.ndarray:
        ps = np.zeros((2, 7))
        for arm, k in ARM_IDS.items():
            if arm in self.previous_sol:
                ps[k] = self.previous_sol[arm]
        return ps

    def _upload_arms(self) -> None:
        for s in self.symbolic_ik_solver.values():
            s._upload()
        self._solver.set_option(_abi.OPT_EULER_ROUNDTRIP,
                                {"auto": _abi.EULER_AUTO, "always": _abi.EULER_ALWAYS, "never": _abi.EULER_NEVER}[self.euler_roundtrip])

    def matrices_to_poses(self, M: Any, identity_shortcut: bool = True) -> torch.Tensor:
        """Batched control_ik.py:212-217: goal matrices ([n,4,4] or [12,n]) -> device poses [6,n]
        (px,py,pz,roll,pitch,yaw), the input layout of SymbolicIK.solve_batch."""
        return self._solver.matrix_to_pose(matrices_to_m12_soa(M, self._solver.device), identity_shortcut=identity_shortcut)

    # One scalar call = one launch and one stream synchronisation: the goal matrix and whatever else the call reads, the
    # joints, the flags and, in continuous mode, the trajectory state sit in ONE pinned host buffer that the device addresses
    # directly (no upload, no download: 38 -> 36 us discrete, 41 -> 40 us continuous).  Layout of the packed buffer, in
    # doubles: [0:19] continuous state, [19:31] goal matrix (12), [31:43] current pose (12), [43:50]
    # current_joints, [50:57] joints out, then 8 bytes: reachable, state, emergency, timed_out.
    _IO_M, _IO_CP, _IO_CJ, _IO_J, _IO_B = 19, 31, 43, 50, 57

    def _scalar_io(self):
        io = getattr(self, "_io", None)
        if io is None:
            import ctypes as C

            nd = self._IO_B + 1
            h = torch.zeros(nd, dtype=torch.float64).pin_memory()
            base = h.data_ptr()
            io = self._io = {
                "h": h, "h_np": h.numpy(), "h_bytes": h.numpy().view(np.uint8),
                "state": C.c_void_p(base), "m_cols": (C.c_void_p * 12)(*[base + 8 * (self._IO_M + k) for k in range(12)]),
                "cp_cols": (C.c_void_p * 12)(*[base + 8 * (self._IO_CP + k) for k in range(12)]),
                "cj": C.c_void_p(base + 8 * self._IO_CJ), "joints": C.c_void_p(base + 8 * self._IO_J),
                "reachable": C.c_void_p(base + 8 * self._IO_B), "code": C.c_void_p(base + 8 * self._IO_B + 1),
                "emergency": C.c_void_p(base + 8 * self._IO_B + 2), "timed_out": C.c_void_p(base + 8 * self._IO_B + 3),
                "pts": np.zeros(2), "ps": np.zeros((2, 7)),
            }
        return io

    @staticmethod
    def _pack_m12(dst: np.ndarray, M: np.ndarray) -> None:
        dst[0:9] = M[:3, :3].reshape(9)
        dst[9:12] = M[:3, 3]

    def _discrete_scalar(self, name: str, M: np.ndarray, current_joints: Any, constrained_mode: str, preferred_theta: float):
        import ctypes as C

        io = self._scalar_io()
        hn = io["h_np"]
        self._pack_m12(hn[self._IO_M: self._IO_M + 12], M)
        hn[self._IO_CJ: self._IO_CJ + 7] = current_joints
        sv = self._solver
        io["ps"][:] = self._previous_sol_2x7()
        self._upload_arms()
        with torch.cuda.device(sv.device):
            sv._bind_stream()
            sv._check(sv.lib.rsik_control_discrete(
                sv._h, 1, io["m_cols"], None, ARM_IDS[name], int(self.nb_search_points), float(preferred_theta),
                _abi.MODES[constrained_mode], io["ps"].ctypes.data_as(C.POINTER(C.c_double)), io["cj"],
                float(self.orbita3D_max_angle), io["joints"], io["reachable"], io["code"], io["emergency"]))
            torch.cuda.current_stream(sv.device).synchronize()
        flags = io["h_bytes"][8 * self._IO_B: 8 * self._IO_B + 4]
        if int(flags[1]) == _abi.STATE_INVALID_INPUT:  # rsik.h "Rows that are not numbers": the scalar drop-in raises like the reference
            raise np.linalg.LinAlgError("SVD did not converge")
        return hn[self._IO_J: self._IO_J + 7].tolist(), bool(flags[0]), STATE_STRINGS[int(flags[1])], int(flags[2])

    # ------------------------------------------------------------------ reference API
    def symbolic_inverse_kinematics(
        self,
        name: str,
        M: npt.NDArray[np.float64],
        control_type: str,
        current_joints: list = [],
        constrained_mode: str = "unconstrained",
        current_pose: npt.NDArray[np.float64] = np.array([]),
        d_theta_max: float = 0.01,
        preferred_theta: float = -4 * np.pi / 6,
    ) -> Tuple[npt.NDArray[np.float64], bool, str]:
        """control_ik.py:162-274."""
        if control_type == "unfreeze":  # clears a latched emergency stop, then behaves like "continuous"
            self.emergency_stop, self.emergency_state, self.init = False, "", True
            self._say(f"{name} Unfreeze", 1.0)
        if self.emergency_stop:  # latched: keep returning the last good solution
            self._say(f"{name} Emergency state: {self.emergency_state}", 1.0)
            return self.previous_sol[name], False, self.emergency_state
        if constrained_mode not in _abi.MODES:
            # the reference leaves interval_limit unbound here (control_ik.py:225-232)
            raise UnboundLocalError("local variable 'interval_limit' referenced before assignment")
        if current_joints == []:
            current_joints = self.previous_sol[name].tolist()
        if len(current_pose) == 0:
            current_pose = self.previous_pose[name]
        if control_type == "continuous" or control_type == "unfreeze":
            ik_joints, is_reachable, state = self._continuous_scalar(
                name, np.asarray(M, dtype=np.float64), current_joints, np.asarray(current_pose, dtype=np.float64),
                constrained_mode, d_theta_max, preferred_theta)
        elif control_type == "discrete":
            M = np.asarray(M, dtype=np.float64)
            if name not in self.symbolic_ik_solver:
                raise KeyError(name)
            ik_joints, is_reachable, state, cause = self._discrete_scalar(
                name, M, np.asarray(current_joints, dtype=np.float64).reshape(7), constrained_mode, preferred_theta)
            if cause:  # control_ik.py:486-495: multiturn_safety_check's messages, one per joint that hit +-6 pi
                self.emergency_stop = True
                self.emergency_state += emergency_messages(cause)
        else:
            raise ValueError(f"Unknown type {control_type}")
        self.previous_pose[name] = M
        return ik_joints, is_reachable, state

    def _continuous_scalar(self, name: str, M: np.ndarray, current_joints: list, current_pose: np.ndarray,
                           constrained_mode: str, d_theta_max: float, preferred_theta: float):
        """control_ik.py:276-407 for one call: the object's previous_theta / previous_sol / init / emergency_stop are
        mirrored into a one-trajectory device state, the step kernel runs, and the attributes are read back."""
        t = time.time()
        timed_out = abs(t - self.last_call_t[name]) > self.call_timeout
        self.last_call_t[name] = t
        import ctypes as C

        io = self._scalar_io()
        hn = io["h_np"]
        hn[0] = self.previous_theta[name]
        has_prev = len(self.previous_sol[name]) == 7
        previous_sol = np.array(self.previous_sol[name], dtype=np.float64) if has_prev else np.zeros(7)
        hn[1:8] = previous_sol
        hn[8] = 1.0 if self.init else 0.0
        hn[9] = 0.0
        hn[10] = 1.0 if has_prev else 0.0
        hn[11:19] = 0.0
        self._pack_m12(hn[self._IO_M: self._IO_M + 12], M)
        self._pack_m12(hn[self._IO_CP: self._IO_CP + 12], current_pose)
        cj = np.asarray(current_joints, dtype=np.float64)
        has_cj = cj.size == 7
        if has_cj:
            hn[self._IO_CJ: self._IO_CJ + 7] = cj.reshape(7)
        io["h_bytes"][8 * self._IO_B + 3] = 1 if timed_out else 0
        io["pts"][:] = [self.preferred_theta.get("r_arm", -4 * np.pi / 6), self.preferred_theta.get("l_arm", -np.pi + 4 * np.pi / 6)]
        sv = self._solver
        self._upload_arms()
        with torch.cuda.device(sv.device):
            sv._bind_stream()
            sv._check(sv.lib.rsik_control_continuous_step(
                sv._h, 1, io["m_cols"], io["cp_cols"], None, ARM_IDS[name], io["timed_out"], float(preferred_theta),
                io["pts"].ctypes.data_as(C.POINTER(C.c_double)), _abi.MODES[constrained_mode], float(d_theta_max),
                io["cj"] if has_cj else None, float(self.orbita3D_max_angle), io["state"], io["joints"], io["reachable"],
                io["code"]))
            torch.cuda.current_stream(sv.device).synchronize()
        back = hn[0:19]
        ik_joints = hn[self._IO_J: self._IO_J + 7].copy()
        flags = io["h_bytes"][8 * self._IO_B: 8 * self._IO_B + 2]
        self.previous_theta[name] = float(back[0])
        self.init = bool(back[8])
        if int(flags[1]) == _abi.STATE_INVALID_INPUT:  # rsik.h "Rows that are not numbers" (previous_sol, init, the latch: untouched)
            raise np.linalg.LinAlgError("SVD did not converge")
        if int(flags[1]) == _abi.STATE_NOT_REACHABLE_NO_LIMITS:
            # control_ik.py:385-387: is_reachable_no_limits failed (a solver with a non-positive projection_margin).  The
            # reference has by now (re)initialised previous_sol / previous_theta if the call timed out, and nothing else.
            self.previous_sol[name] = back[1:8].copy()
            print(f"{name} Pose not reachable, this has to be fixed by projecting far poses to reachable sphere")
            raise RuntimeError("Pose not reachable in symbolic IK. We crash on purpose while we are on the debug sessions.")
        if back[9] != 0.0:  # control_ik.py:486-495, 396-401: the reference's own diagnostics for what tripped
            self.emergency_stop = True
            # continuity_check prints the previous_sol the step was checked against: after a (re)initialisation that is
            # current_joints (control_ik.py:312), i.e. the state rows the kernel hands back
            self.emergency_state += emergency_messages(int(back[11]), previous_joints=back[1:8].copy(), joints=back[12:19].copy())
        if not self.emergency_stop:
            self.previous_sol[name] = copy.deepcopy(ik_joints)
        else:
            self.previous_sol[name] = back[1:8].copy()
        return ik_joints, bool(flags[0]), STATE_STRINGS[int(flags[1])]

    def emergency_report(self, cont_state: torch.Tensor) -> Dict[int, str]:
        """For a batch state (new_continuous_state): trajectory index -> the text the reference would have put in
        emergency_state when that trajectory's emergency stop tripped (cause bits + rejected joints, state rows 11-18)."""
        st = cont_state.detach().cpu().numpy()
        out: Dict[int, str] = {}
        for i in np.nonzero(st[9] != 0.0)[0]:
            out[int(i)] = emergency_messages(int(st[11, i]), previous_joints=st[1:8, i], joints=st[12:19, i])
        return out

    # ------------------------------------------------------------------ MI355X-native batch API
    def new_continuous_state(self, name: Any, n: int) -> torch.Tensor:
        """Per-trajectory state [RSIK_CONT_STATE_ROWS, n] initialised like a freshly constructed ControlIK (control_ik.py:133-160):
        previous_theta / previous_sol of the arm(s), init = True, no emergency stop."""
        st = self._solver.new_continuous_state(n)
        if isinstance(name, str):
            arm_ids = np.full(n, ARM_IDS[name], dtype=np.int64)
        else:
            arm_ids = (name.cpu().numpy() if isinstance(name, torch.Tensor) else np.asarray(name)).astype(np.int64)
        names = ["r_arm", "l_arm"]
        host = np.zeros((_abi.CONT_STATE_ROWS, n))
        for k in (0, 1):
            m = arm_ids == k
            if m.any():
                host[0, m] = self.previous_theta[names[k]]
                host[1:8, m] = np.asarray(self.previous_sol[names[k]], dtype=np.float64).reshape(7, 1)
        host[8] = 1.0
        host[10] = 1.0
        st.copy_(torch.as_tensor(host))
        return st

    def symbolic_inverse_kinematics_continuous_batch(
        self,
        name: Any,
        M: Any,
        cont_state: torch.Tensor,
        timed_out: Any = None,
        current_joints: Any = None,
        current_pose: Any = None,
        constrained_mode: str = "unconstrained",
        d_theta_max: float = 0.01,
        preferred_theta: float = -4 * np.pi / 6,
        out: Optional[Dict[str, torch.Tensor]] = None,
    ) -> Dict[str, torch.Tensor]:
        """Continuous-mode IK, one control step for n independent trajectories (state carried in `cont_state`,
        updated in place).  `timed_out` [n] uint8 replaces the reference's 0.2 s wall-clock test; pass 1 on the first
        step of a trajectory to reproduce the reference's start-up (re-initialisation from current_joints /
        current_pose, control_ik.py:296-325)."""
        if constrained_mode not in _abi.MODES:
            raise UnboundLocalError("local variable 'interval_limit' referenced before assignment")
        dev = self._solver.device
        m12 = matrices_to_m12_soa(M, dev)
        cp = None if current_pose is None else matrices_to_m12_soa(current_pose, dev)
        arm_t, arm_uniform = None, 0
        if isinstance(name, str):
            arm_uniform = ARM_IDS[name]
        else:
            arm_t = name
        pts = [self.preferred_theta.get("r_arm", -4 * np.pi / 6), self.preferred_theta.get("l_arm", -np.pi + 4 * np.pi / 6)]
        self._upload_arms()
        return self._solver.control_continuous_step(
            m12, cont_state, pts, arm=arm_t, arm_uniform=arm_uniform, timed_out=timed_out,
            preferred_theta=float(preferred_theta), constrained_mode=_abi.MODES[constrained_mode],
            d_theta_max=float(d_theta_max), current_joints=current_joints, current_pose_m12=cp,
            orbita3d_max_angle=float(self.orbita3D_max_angle), out=out)

    def symbolic_inverse_kinematics_batch(
        self,
        name: Any,
        M: Any,
        constrained_mode: str = "unconstrained",
        current_joints: Any = None,
        preferred_theta: float = -4 * np.pi / 6,
        out: Optional[Dict[str, torch.Tensor]] = None,
        plan_only: bool = False,
    ) -> Dict[str, torch.Tensor]:
        """Discrete-mode IK for a batch of goal matrices.

        name: "r_arm" / "l_arm" for a single-arm batch, or a uint8 tensor/array [n] of arm ids (0 = r, 1 = l).
        M: [n,4,4] or packed SoA [12,n].  current_joints: [n,7] or None (=> previous_sol of the pose's arm).
        Returns device tensors joints [n,7], reachable [n], state [n], emergency [n].
        """
        if constrained_mode not in _abi.MODES:
            raise UnboundLocalError("local variable 'interval_limit' referenced before assignment")
        m12 = matrices_to_m12_soa(M, self._solver.device)
        arm_t, arm_uniform = None, 0
        if isinstance(name, str):
            if name not in self.symbolic_ik_solver:
                raise KeyError(name)
            arm_uniform = ARM_IDS[name]
        else:
            arm_t = name
        self._upload_arms()
        return self._solver.control_discrete(
            m12, arm=arm_t, arm_uniform=arm_uniform, nb_search_points=int(self.nb_search_points),
            preferred_theta=float(preferred_theta), constrained_mode=_abi.MODES[constrained_mode],
            previous_sol=self._previous_sol_2x7(), current_joints=current_joints,
            orbita3d_max_angle=float(self.orbita3D_max_angle), out=out, plan_only=plan_only)

    def run_continuous_trajectories(
        self,
        name: Any,
        M_steps: Any,
        cont_state: torch.Tensor,
        first_step_timed_out: bool = True,
        current_joints: Any = None,
        current_pose: Any = None,
        constrained_mode: str = "unconstrained",
        d_theta_max: float = 0.01,
        preferred_theta: float = -4 * np.pi / 6,
        out: Optional[Dict[str, torch.Tensor]] = None,
        goals_resident: Optional[bool] = None,
    ) -> Dict[str, torch.Tensor]:
        """All steps of n parallel trajectories with one host call.  M_steps: [n_steps, n, 4, 4] or packed
        [n_steps, 12, n].  Returns joints [n_steps, n, 7], reachable / state [n_steps, n] (a dict with the attribute `run_form`
        beside its keys: how the library issued the run (_abi.CONT_FORM_*; CONT_FORM_STEPS_NO_LIMITS_CAN_FAIL = a launch per control step because this arm's projection
        margin lets is_reachable_no_limits fail, symbolic_ik.py:343-345 / control_ik.py:385-387 — correct, 10-30 x slower)).
        `goals_resident=True`: a promise that lets consecutive runs of one shape overlap (the prepare phase of this run beside the
        tail of the one before): M_steps was complete on the device before the previous run of this object was issued, and nothing
        queued since uses this run's `out` buffers — include/rsik.h, RSIK_OPT_CONT_GOALS_RESIDENT."""
        if constrained_mode not in _abi.MODES:
            raise UnboundLocalError("local variable 'interval_limit' referenced before assignment")
        dev = self._solver.device
        t = M_steps if isinstance(M_steps, torch.Tensor) else torch.as_tensor(np.asarray(M_steps, dtype=np.float64))
        t = t.to(device=dev, dtype=torch.float64)
        if t.dim() == 4 and tuple(t.shape[2:]) == (4, 4):
            rot = t[:, :, :3, :3].reshape(t.shape[0], t.shape[1], 9)
            t = torch.cat([rot, t[:, :, :3, 3]], dim=2).permute(0, 2, 1)
        m12_steps = t.contiguous()
        cp = None if current_pose is None else matrices_to_m12_soa(current_pose, dev)
        arm_t, arm_uniform = (None, ARM_IDS[name]) if isinstance(name, str) else (name, 0)
        pts = [self.preferred_theta.get("r_arm", -4 * np.pi / 6), self.preferred_theta.get("l_arm", -np.pi + 4 * np.pi / 6)]
        self._upload_arms()
        return self._solver.control_continuous_run(
            m12_steps, cont_state, pts, arm=arm_t, arm_uniform=arm_uniform, first_step_timed_out=first_step_timed_out,
            preferred_theta=float(preferred_theta), constrained_mode=_abi.MODES[constrained_mode],
            d_theta_max=float(d_theta_max), current_joints=current_joints, current_pose_m12=cp,
            orbita3d_max_angle=float(self.orbita3D_max_angle), out=out, goals_resident=goals_resident)

    def capture_continuous_trajectories(
        self,
        name: Any,
        M_steps: Any,
        cont_state: torch.Tensor,
        first_step_timed_out: bool = True,
        current_joints: Any = None,
        current_pose: Any = None,
        constrained_mode: str = "unconstrained",
        d_theta_max: float = 0.01,
        preferred_theta: float = -4 * np.pi / 6,
        out: Optional[Dict[str, torch.Tensor]] = None,
    ):
        """run_continuous_trajectories recorded once into a hipGraph: returns (graph, out) — `graph.replay()` re-runs the whole
        run on the buffers it was captured with (M_steps, cont_state and `out` are read / written in place: refill M_steps and
        reset or keep cont_state between replays as the caller needs).  What a caller that solves batch after batch of the same
        shape should use: a replay costs the host one call and is 1-3 % faster than issuing the pipeline's launches one by one
        (4096 x 1000: 0.363-0.373 against 0.369-0.390 ms, docs/experiments.md A.6).  The capture needs nothing created: the
        workspace, side streams and events are reserved first, and every per-trajectory argument (a tensor of arm ids,
        current_joints, current_pose) is brought to the device BEFORE the capture begins — a host array converted inside it would
        be a copy on the capturing stream from memory the graph does not own.  Those device copies are kept alive on the returned
        graph object (`graph.rsik_inputs`): refill them in place to replay with other values.  The scalars (first_step_timed_out,
        constrained_mode, d_theta_max, preferred_theta, this object's per-arm preferred_theta and the arm parameters as uploaded
        now) are frozen into the graph: a run with other values is another capture."""
        if constrained_mode not in _abi.MODES:
            raise UnboundLocalError("local variable 'interval_limit' referenced before assignment")
        dev = self._solver.device
        t = M_steps if isinstance(M_steps, torch.Tensor) else torch.as_tensor(np.asarray(M_steps, dtype=np.float64))
        if not (t.dim() == 3 and t.shape[1] == 12 and t.dtype == torch.float64 and t.device == dev and t.is_contiguous()):
            raise ValueError("capture_continuous_trajectories: M_steps must be a contiguous float64 [n_steps, 12, n] tensor on the device "
                             "(the graph reads it in place)")
        n_steps, _, n = (int(v) for v in t.shape)
        held: Dict[str, torch.Tensor] = {}
        if not isinstance(name, str):
            name = held["arm"] = self._solver._dev_u8(name, n, "arm")
        if current_joints is not None:
            current_joints = held["current_joints"] = self._solver._dev_f64(current_joints, (n, 7), "current_joints")
        if current_pose is not None:
            current_pose = held["current_pose"] = matrices_to_m12_soa(current_pose, dev)
        if out is None:
            out = {"joints": torch.empty((n_steps, n, 7), dtype=torch.float64, device=dev),
                   "reachable": torch.empty((n_steps, n), dtype=torch.uint8, device=dev),
                   "state": torch.empty((n_steps, n), dtype=torch.uint8, device=dev)}
        else:
            for key, shape, dt in (("joints", (n_steps, n, 7), torch.float64), ("reachable", (n_steps, n), torch.uint8),
                                   ("state", (n_steps, n), torch.uint8)):
                o = out.get(key)
                if o is None or tuple(o.shape) != shape or o.dtype != dt or o.device != dev or not o.is_contiguous():
                    raise ValueError(f"capture_continuous_trajectories: out[{key!r}] must be a contiguous {dt} {list(shape)} tensor on the device")
        self._upload_arms()
        self._solver.control_continuous_reserve(n, n_steps)
        torch.cuda.synchronize(dev)  # the conversions above are done before anything is recorded
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                self.run_continuous_trajectories(name, t, cont_state, first_step_timed_out=first_step_timed_out,
                                                 current_joints=current_joints, current_pose=current_pose,
                                                 constrained_mode=constrained_mode, d_theta_max=d_theta_max,
                                                 preferred_theta=preferred_theta, out=out)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph.rsik_inputs = held
        return graph, out

