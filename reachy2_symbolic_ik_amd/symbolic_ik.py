"""SymbolicIK — drop-in for reachy2_symbolic_ik.symbolic_ik.SymbolicIK (symbolic_ik.py:25-863) on MI355X.

Same constructor signature, public attributes, return tuples and state strings as the reference;
every solve runs in the HIP kernels behind include/rsik.h (no CPU path).  The scalar methods
(`is_reachable`, `get_joints`, ...) are batch-of-one calls that keep the reference's per-instance
state semantics in a one-row device state array; the `*_batch` methods are the
MI355X-native interface for pose batches laid out SoA in HBM.
"""
from __future__ import annotations

from typing import Any, Dict, Optional, Sequence, Tuple

import numpy as np
import numpy.typing as npt
import torch

from . import _abi
from .backend import HipSolver
from .constants import ARM_IDS, STATE_STRINGS, ArmGeometry, default_ik_parameters

_THETA_POLICIES = {"interval0": _abi.THETA_INTERVAL0, "explicit": _abi.THETA_EXPLICIT, "fraction": _abi.THETA_FRACTION,
                   "none": _abi.THETA_NONE}


def poses_to_soa(poses: Any, device: torch.device) -> torch.Tensor:
    """[n,2,3] (position, xyz-euler) or [6,n] -> [6,n] float64 on `device` with unit-stride rows (a column slice of a
    larger SoA batch is passed through as a view: the ABI takes one pointer per column array)."""
    t = poses if isinstance(poses, torch.Tensor) else torch.as_tensor(np.asarray(poses, dtype=np.float64))
    t = t.to(device=device, dtype=torch.float64)
    if t.dim() == 3 and tuple(t.shape[1:]) == (2, 3):
        t = t.reshape(t.shape[0], 6).t()
    elif not (t.dim() == 2 and t.shape[0] == 6):
        raise ValueError("poses must have shape [n,2,3] or [6,n]")
    return t if (t.shape[1] <= 1 or t.stride(1) == 1) else t.contiguous()


class SymbolicIK:
    def __init__(
        self,
        arm: str = "r_arm",
        ik_parameters: Dict[str, Any] = {},
        elbow_limit: int = 127,
        wrist_limit: np.float64 = np.float64(42.5),
        projection_margin: float = 1e-8,
        backward_limit: float = 0.02,
        normal_vector_margin: float = 1e-7,
        singularity_offset: float = 0.03,
        singularity_limit_coeff: float = 1.0,
        device: Any = None,
        solver: Optional[HipSolver] = None,
    ) -> None:
        if ik_parameters == {}:
            print("Using default parameters")
            ik_parameters = default_ik_parameters()
        self._geom = ArmGeometry(arm, ik_parameters, elbow_limit, wrist_limit, projection_margin, backward_limit,
                                 normal_vector_margin, singularity_offset, singularity_limit_coeff)
        g = self._geom
        # public attributes of the reference (symbolic_ik.py:55-83)
        self.arm = arm
        self.shoulder_position = g.shoulder_position
        self.shoulder_orientation_offset = g.shoulder_orientation_offset
        self.upper_arm_size = g.upper_arm_size
        self.forearm_size = g.forearm_size
        self.tip_position = g.tip_position
        self.gripper_size = g.gripper_size
        self.max_arm_length = g.max_arm_length
        self.torso_pose = np.array([0.0, 0.0, 0.0])
        self.projection_margin = projection_margin
        self.normal_vector_margin = normal_vector_margin
        self.backward_limit = backward_limit
        self.elbow_limit = elbow_limit
        self.shoulder_wrist_min_distance = g.shoulder_wrist_min_distance
        self.wrist_limit = wrist_limit
        self.singularity_offset = singularity_offset
        self.singularity_limit_coeff = singularity_limit_coeff
        self.elbow_singularity_position = g.elbow_singularity_position
        self.wrist_singularity_position = g.wrist_singularity_position

        self.arm_id = ARM_IDS[arm]
        self.consts = g.pack()
        self._solver = solver if solver is not None else HipSolver(device)
        self._solver.set_arm(self.arm_id, self.consts)

    # ------------------------------------------------------------------ scalar drop-in API
    @property
    def solver(self) -> HipSolver:
        return self._solver

    def _upload(self) -> None:
        # several SymbolicIK objects may share one context; make sure *this* arm's constants are current
        self._solver.set_arm(self.arm_id, self.consts)

    # One scalar call = one launch and one stream synchronisation: the kernel reads its arguments from, and keeps this
    # object's row of solver state (which carries the call's results, include/rsik.h RSIK_SOLVER_STATE_STRIDE) in,
    # pinned host memory, which the device addresses directly — no upload, no download (is_reachable 32 -> 23 us,
    # get_elbow_position 27 -> 20 us, get_joints 30 -> 27 us per call against staging copies on both sides).
    def _scalar_io(self):
        io = getattr(self, "_io", None)
        if io is None:
            import ctypes as C

            h_in = torch.empty(8, dtype=torch.float64).pin_memory()
            h_state = torch.zeros(_abi.SOLVER_STATE_STRIDE, dtype=torch.float64).pin_memory()  # one instance = one row
            h_elbow = torch.empty(3, dtype=torch.float64).pin_memory()
            base = h_in.data_ptr()
            io = self._io = {
                "h_in": h_in, "h_in_np": h_in.numpy(), "h_state": h_state, "h_state_np": h_state.numpy(),
                "h_elbow": h_elbow, "h_elbow_np": h_elbow.numpy(),
                "cols": (C.c_void_p * 6)(*[base + 8 * k for k in range(6)]),
                "theta": C.c_void_p(base), "prev": C.c_void_p(base + 8), "state": C.c_void_p(h_state.data_ptr()),
                "elbow": C.c_void_p(h_elbow.data_ptr()),
            }
        return io

    def _finish(self, io) -> np.ndarray:
        torch.cuda.current_stream(self._solver.device).synchronize()
        s = io["h_state_np"]
        self.goal_pose = np.array([s[0:3], s[3:6]])
        self.wrist_position = s[6:9].copy()
        self.intersection_circle = (s[9:12].copy(), float(s[12]), s[13:16].copy())
        return s

    def _reach_scalar(self, goal_pose: Any, no_limits: bool):
        io = self._scalar_io()
        io["h_in_np"][0:3] = goal_pose[0]
        io["h_in_np"][3:6] = goal_pose[1]
        sv = self._solver
        self._upload()
        with torch.cuda.device(sv.device):
            sv._bind_stream()
            sv._check(sv.lib.rsik_reach_state(sv._h, 1, io["cols"], None, self.arm_id, 1 if no_limits else 0, io["state"],
                                              None, None, None))
            s = self._finish(io)
        if int(s[23]) == _abi.STATE_INVALID_INPUT:
            # include/rsik.h "Rows that are not numbers": the batch API reports the row; the scalar drop-in does what the reference
            # does with a NaN in the pose (symbolic_ik.py:580, np.linalg.lstsq on a system full of NaN) — the solver object is untouched
            raise np.linalg.LinAlgError("SVD did not converge in Linear Least Squares")
        return bool(s[22] != 0.0), s[20:22].copy(), int(s[23])

    def is_reachable(self, goal_pose: npt.NDArray[np.float64]) -> Tuple[bool, npt.NDArray[np.float64], Optional[Any], str]:
        """symbolic_ik.py:121-282."""
        ok, interval, code = self._reach_scalar(goal_pose, no_limits=False)
        if not ok:
            return False, np.array([]), None, STATE_STRINGS[code]
        return True, interval, self.get_joints, STATE_STRINGS[code]

    def is_reachable_no_limits(self, goal_pose: npt.NDArray[np.float64]) -> Tuple[bool, npt.NDArray[np.float64], Optional[Any]]:
        """symbolic_ik.py:85-119."""
        ok, interval, _ = self._reach_scalar(goal_pose, no_limits=True)
        if not ok:
            return False, np.array([]), None
        return True, np.array([-np.pi, np.pi]), self.get_joints

    def get_joints(
        self, theta: float, previous_joints: Sequence[float] = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    ) -> Tuple[npt.NDArray[np.float64], npt.NDArray[np.float64]]:
        """symbolic_ik.py:697-863 — reads and (when the elbow projection fires) updates the state left by the
        last is_reachable*() call, exactly like the reference's bound method."""
        io = self._scalar_io()
        io["h_in_np"][0] = theta
        io["h_in_np"][1:8] = previous_joints
        sv = self._solver
        self._upload()
        with torch.cuda.device(sv.device):
            sv._bind_stream()
            sv._check(sv.lib.rsik_joints_from_state(sv._h, 1, io["state"], None, self.arm_id, io["theta"], io["prev"], None, None))
            s = self._finish(io)
        joints = s[24:31].copy()
        if s[19] != 0.0:  # Q2: 3 components after a projection, [x, y, z, 1] otherwise
            self.elbow_position = s[16:19].copy()
        else:
            self.elbow_position = np.array([s[16], s[17], s[18], 1.0])
        return joints, self.elbow_position

    def get_elbow_position(self, theta: float) -> npt.NDArray[np.float64]:
        """symbolic_ik.py:684-695."""
        io = self._scalar_io()
        io["h_in_np"][0] = theta
        sv = self._solver
        with torch.cuda.device(sv.device):
            sv._bind_stream()
            sv._check(sv.lib.rsik_elbow_from_state(sv._h, 1, io["state"], io["theta"], io["elbow"]))
            torch.cuda.current_stream(sv.device).synchronize()
        e = io["h_elbow_np"]
        return np.array([e[0], e[1], e[2], 1.0])

    # ---- the stages of is_reachable as public methods (the reference's own harness times them one by one,
    # src/benchmark/ik_benchmarks.py:36-130): each is one rsik_stage launch on the operands it is given — the fused kernels never
    # form these intermediates.  `self.wrist_position` is read where the reference reads it (a caller may assign it).
    def _stage(self, op: int, *operands: Any) -> np.ndarray:
        need_in, need_out = _abi.STAGE_ROW[op]
        io = getattr(self, "_stage_io", None)
        if io is None:
            io = self._stage_io = {"in": torch.empty((1, _abi.STAGE_IN_MAX), dtype=torch.float64).pin_memory(),
                                   "out": torch.empty((1, _abi.STAGE_OUT_MAX), dtype=torch.float64).pin_memory()}
            io["in_np"], io["out_np"] = io["in"].numpy(), io["out"].numpy()
        flat = np.concatenate([np.asarray(a, dtype=np.float64).reshape(-1) for a in operands])
        if flat.size != need_in:
            raise ValueError(f"stage {op} takes {need_in} numbers, got {flat.size}")
        io["in_np"][0, :need_in] = flat
        sv = self._solver
        self._upload()
        with torch.cuda.device(sv.device):
            sv._bind_stream()
            sv._check(sv.lib.rsik_stage(sv._h, int(op), 1, self.arm_id, io["in"].data_ptr(), _abi.STAGE_IN_MAX, io["out"].data_ptr(), _abi.STAGE_OUT_MAX))
            torch.cuda.current_stream(sv.device).synchronize()
        return io["out_np"][0, :need_out].copy()

    def is_pose_in_robot_reach(self, goal_pose: npt.NDArray[np.float64]) -> Tuple[bool, npt.NDArray[np.float64], str]:
        """symbolic_ik.py:284-307."""
        o = self._stage(_abi.STAGE_POSE_IN_REACH, goal_pose[0], goal_pose[1])
        code = int(o[4])
        return bool(o[0] != 0.0), np.array([o[1:4], np.asarray(goal_pose[1], dtype=np.float64)]), STATE_STRINGS[code]

    def get_wrist_position(self, goal_pose: npt.NDArray[np.float64]) -> npt.NDArray[np.float64]:
        """symbolic_ik.py:418-425."""
        return self._stage(_abi.STAGE_WRIST_POSITION, goal_pose[0], goal_pose[1])

    def get_limitation_wrist_circle(self, goal_pose: npt.NDArray[np.float64]) -> Tuple[npt.NDArray[np.float64], float, npt.NDArray[np.float64]]:
        """symbolic_ik.py:401-416 (reads self.wrist_position)."""
        o = self._stage(_abi.STAGE_LIMITATION_CIRCLE, self.wrist_position, goal_pose[0])
        return o[0:3], float(o[3]), o[4:7]

    def get_intersection_circle(self, goal_pose: npt.NDArray[np.float64]) -> Optional[Tuple[npt.NDArray[np.float64], float, npt.NDArray[np.float64]]]:
        """symbolic_ik.py:366-399 (reads self.wrist_position; the argument is not looked at, as in the reference)."""
        o = self._stage(_abi.STAGE_INTERSECTION_CIRCLE, self.wrist_position)
        return None if o[0] == 0.0 else (o[1:4], float(o[4]), o[5:8])

    def are_circles_linked(self, intersection_circle: Any, limitation_wrist_circle: Any) -> npt.NDArray[np.float64]:
        """symbolic_ik.py:427-509: the interval of valid elbow angles, or an empty array (reads self.wrist_position)."""
        o = self._stage(_abi.STAGE_CIRCLES_LINKED, self.wrist_position, intersection_circle[0], [intersection_circle[1]], intersection_circle[2],
                        limitation_wrist_circle[0], [limitation_wrist_circle[1]], limitation_wrist_circle[2])
        return np.array([]) if o[0] == 0.0 else o[1:3]

    def points_of_nearest_approach(self, p1: Any, V_torso_normal1: Any, p2: Any, V_torso_normal2: Any) -> Tuple[npt.NDArray[np.float64], npt.NDArray[np.float64]]:
        """symbolic_ik.py:588-606: (a point of, the direction of) the line in which the two circles' planes meet."""
        o = self._stage(_abi.STAGE_NEAREST_APPROACH, p1, V_torso_normal1, p2, V_torso_normal2)
        return (o[1:4] if o[0] != 0.0 else np.array([])), o[4:7]

    def intersection_circle_line_3d_vd(self, center: Any, radius: float, direction: Any, point_on_line: Any) -> Optional[npt.NDArray[np.float64]]:
        """symbolic_ik.py:608-645."""
        o = self._stage(_abi.STAGE_CIRCLE_LINE, center, [radius], direction, point_on_line)
        k = int(o[0])
        return None if k == 0 else (np.array([o[1:4]]) if k == 1 else np.vstack((o[1:4], o[4:7])))

    # ------------------------------------------------------------------ batched API (MI355X-native)
    def solve_batch(
        self,
        poses: Any,
        theta: Any = "interval0",
        previous_joints: Optional[Sequence[float]] = None,
        want_elbow: bool = True,
        out: Optional[Dict[str, torch.Tensor]] = None,
        plan_only: bool = False,
    ) -> Dict[str, torch.Tensor]:
        """Fused is_reachable + get_joints for a batch of poses of this arm.

        poses: [n,2,3] or SoA [6,n].  theta: "interval0" (the reference's README/benchmark convention),
        "none" (reachability only), ("explicit", tensor[n]) or ("fraction", tensor[n]).
        Returns device tensors: joints [n,7], interval [n,2], elbow [n,3], reachable [n] u8, state [n] u8.
        """
        soa = poses_to_soa(poses, self._solver.device)
        theta_in = None
        if isinstance(theta, str):
            policy = _THETA_POLICIES[theta]
        else:
            policy = _THETA_POLICIES[theta[0]]
            theta_in = theta[1]
        self._upload()
        return self._solver.solve(soa, arm_uniform=self.arm_id, theta_policy=policy, theta_in=theta_in,
                                  previous_joints=previous_joints, want_elbow=want_elbow, out=out, plan_only=plan_only)

    def is_reachable_batch(self, poses: Any) -> Dict[str, torch.Tensor]:
        return self.solve_batch(poses, theta="none")

    def solve_batch_host(self, poses_soa_host: Any, chunk: Optional[int] = None, slots: int = 3,
                         out: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """solve_batch for a batch that lives in HOST memory (what a caller of the reference's NumPy API has).
        poses_soa_host: [6,n] float64 (pinned memory gives the full PCIe rate: torch.Tensor.pin_memory()).  Returns HOST
        tensors (pinned): joints [n,7], interval [n,2], reachable [n], state [n] (theta = interval[0]).

        chunk = None: one upload, one launch, one download.  With a chunk size the batch is cut into chunks whose
        upload / solve / download are queued round-robin on `slots` HIP streams (device buffers of `chunk` poses instead
        of n).  Measured on MI355X (scripts/host_pipeline.py, 4 M poses, pinned): 0.455 G solves/s either way — the
        122 B/pose cross PCIe at 55 GB/s in total, uploads and downloads do not overlap each other on this platform, and
        the kernel is 60x faster than the link; chunks of 1 M / 256 k / 64 k poses cost 6 / 17 / 35 % in launch overhead."""
        dev = self._solver.device
        src = poses_soa_host if isinstance(poses_soa_host, torch.Tensor) else torch.as_tensor(np.asarray(poses_soa_host, dtype=np.float64))
        if src.dim() != 2 or src.shape[0] != 6 or src.dtype != torch.float64 or src.is_cuda:
            raise ValueError("poses_soa_host must be a host float64 tensor/array of shape [6, n]")
        n = int(src.shape[1])
        if out is None:
            out = {
                "joints": torch.empty((n, 7), dtype=torch.float64).pin_memory(),
                "interval": torch.empty((n, 2), dtype=torch.float64).pin_memory(),
                "reachable": torch.empty((n,), dtype=torch.uint8).pin_memory(),
                "state": torch.empty((n,), dtype=torch.uint8).pin_memory(),
            }
        if n == 0:
            return out
        chunk = n if chunk is None else max(1, min(int(chunk), n))
        self._upload()
        cache = getattr(self, "_host_pipeline", None)
        if cache is None or cache["chunk"] != chunk or len(cache["slots"]) != slots:
            cache = {"chunk": chunk, "slots": [
                {"stream": torch.cuda.Stream(device=dev),
                 "in": torch.empty((6, chunk), dtype=torch.float64, device=dev),
                 "joints": torch.empty((chunk, 7), dtype=torch.float64, device=dev),
                 "interval": torch.empty((chunk, 2), dtype=torch.float64, device=dev),
                 "reachable": torch.empty((chunk,), dtype=torch.uint8, device=dev),
                 "state": torch.empty((chunk,), dtype=torch.uint8, device=dev)} for _ in range(slots)]}
            self._host_pipeline = cache
        start = torch.cuda.current_stream(dev).record_event()
        for k, a in enumerate(range(0, n, chunk)):
            b = min(a + chunk, n)
            m = b - a
            sl = cache["slots"][k % slots]
            with torch.cuda.stream(sl["stream"]):
                if k < slots:
                    sl["stream"].wait_event(start)
                for c in range(6):  # one contiguous copy per column (a strided [6, m] copy is staged through pageable memory)
                    sl["in"][c, :m].copy_(src[c, a:b], non_blocking=True)
                dev_out = {key: sl[key][:m] for key in ("joints", "interval", "reachable", "state")}
                self._solver.solve(sl["in"][:, :m] if m == chunk else sl["in"][:, :m].contiguous(), arm_uniform=self.arm_id,
                                   theta_policy=_THETA_POLICIES["interval0"], want_elbow=False, out=dev_out)
                for key in ("joints", "interval", "reachable", "state"):
                    out[key][a:b].copy_(dev_out[key], non_blocking=True)
        for sl in cache["slots"]:
            torch.cuda.current_stream(dev).wait_stream(sl["stream"])
        torch.cuda.current_stream(dev).synchronize()
        return out

    def forward_kinematics_batch(self, joints: Any):
        """joints [n,7] -> (goal position [n,3], goal rotation [n,3,3]): the chain get_joints inverts
        (symbolic_ik.py:728-848).  The reference itself has no FK; this is the on-device self-check of SURVEY 8 f-4."""
        self._upload()
        j = torch.as_tensor(joints, dtype=torch.float64).to(self._solver.device)
        return self._solver.forward_kinematics(j.reshape(-1, 7).contiguous(), arm_uniform=self.arm_id)

    def fk_residual_batch(self, poses: Any, joints: Any) -> torch.Tensor:
        """err [n,2] = (position error m, rotation error rad) of FK(joints) against the poses they were solved for."""
        self._upload()
        soa = poses_to_soa(poses, self._solver.device)
        j = torch.as_tensor(joints, dtype=torch.float64).to(self._solver.device)
        return self._solver.fk_residual(soa, j.reshape(-1, 7).contiguous(), arm_uniform=self.arm_id)

    @staticmethod
    def state_strings(codes: Any) -> list:
        c = codes.cpu().numpy() if isinstance(codes, torch.Tensor) else np.asarray(codes)
        return [STATE_STRINGS[int(k)] for k in c]


class DualArmIK:
    """Mixed r/l batches in one launch (BASELINE config 4): two SymbolicIK objects sharing one device context, the arm
    of every pose given by a uint8 array (0 = r_arm, 1 = l_arm)."""

    def __init__(self, device: Any = None, solver: Optional[HipSolver] = None, **symbolic_ik_kwargs: Any) -> None:
        self._solver = solver if solver is not None else HipSolver(device)
        self.r_arm = SymbolicIK("r_arm", solver=self._solver, **symbolic_ik_kwargs)
        self.l_arm = SymbolicIK("l_arm", solver=self._solver, **symbolic_ik_kwargs)

    @property
    def solver(self) -> HipSolver:
        return self._solver

    def solve_batch(self, arm_ids: Any, poses: Any, theta: Any = "interval0", previous_joints: Optional[Sequence[float]] = None,
                    want_elbow: bool = True, out: Optional[Dict[str, torch.Tensor]] = None,
                    plan_only: bool = False) -> Dict[str, torch.Tensor]:
        soa = poses_to_soa(poses, self._solver.device)
        theta_in = None
        if isinstance(theta, str):
            policy = _THETA_POLICIES[theta]
        else:
            policy = _THETA_POLICIES[theta[0]]
            theta_in = theta[1]
        self.r_arm._upload()
        self.l_arm._upload()
        return self._solver.solve(soa, arm=arm_ids, theta_policy=policy, theta_in=theta_in, previous_joints=previous_joints,
                                  want_elbow=want_elbow, out=out, plan_only=plan_only)
