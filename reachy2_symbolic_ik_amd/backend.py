"""Device-side driver: owns one rsik context per GPU and hands torch-owned HBM buffers to the C ABI.

PyTorch is used for device memory, streams and torch.distributed only; all arithmetic happens in
the hand-written HIP kernels behind include/rsik.h.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, Optional, Sequence

import numpy as np
import torch

from . import _abi
from .constants import ARM_CONSTS_COUNT

_F64 = torch.float64
_U8 = torch.uint8


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class ContinuousRunResult(dict):
    """What control_continuous_run returns: the dict of output tensors (joints, reachable, state) with one attribute beside it,
    `run_form` = how the library issued the run (_abi.CONT_FORM_*; `run_form_name` in words)."""

    run_form: int = _abi.CONT_FORM_NONE

    @property
    def run_form_name(self) -> str:
        return _abi.CONT_FORM_NAMES.get(self.run_form, str(self.run_form))


class HipSolver:
    """One context = one GPU.  Methods enqueue on torch's current stream and return torch tensors."""

    def __init__(self, device: int | torch.device | None = None) -> None:
        self.lib = _abi.load()
        if self.lib.rsik_device_count() <= 0 or not torch.cuda.is_available():
            raise RuntimeError(
                "reachy2_symbolic_ik_amd: no MI355X/HIP device visible — this package has no CPU fallback"
            )
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", index)
        h = C.c_void_p()
        rc = self.lib.rsik_create(index, C.byref(h))
        if rc != _abi.RSIK_OK:
            raise _abi.RsikError(rc, (self.lib.rsik_last_error(None) or b"").decode())
        self._h = h
        self._arms_set = [False, False]
        self._arm_blocks = [None, None]
        self._arm_gen = 0  # bumped whenever an arm's constants really change: plans made before are then stale

    def close(self) -> None:
        if getattr(self, "_h", None):
            self.lib.rsik_destroy(self._h)
            self._h = None

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int) -> None:
        if rc != _abi.RSIK_OK:
            raise _abi.RsikError(rc, (self.lib.rsik_last_error(self._h) or b"").decode())

    def _bind_stream(self) -> None:
        s = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.rsik_set_stream(self._h, C.c_void_p(s)))

    def set_arm(self, arm_id: int, consts: np.ndarray) -> None:
        c = np.ascontiguousarray(consts, dtype=np.float64)
        if c.shape != (ARM_CONSTS_COUNT,):
            raise ValueError(f"expected {ARM_CONSTS_COUNT} constants, got {c.shape}")
        old = self._arm_blocks[arm_id]
        if old is not None and old.tobytes() == c.tobytes():
            return
        self._check(self.lib.rsik_set_arm(self._h, int(arm_id), c.ctypes.data_as(C.POINTER(C.c_double)), c.size))
        self._arms_set[arm_id] = True
        self._arm_blocks[arm_id] = c.copy()
        self._arm_gen += 1

    def synchronize(self) -> None:
        """Waits for the work on the CURRENT torch stream of this device (the stream every non-planned call uses)."""
        with torch.cuda.device(self.device):
            self._bind_stream()  # (a planned launch may have left the context on another, possibly destroyed, stream)
            self._check(self.lib.rsik_sync(self._h))

    def control_continuous_reserve(self, n: int, n_steps: int) -> None:
        """rsik_control_continuous_reserve: workspace, side streams and events of a control_continuous_run(n, n_steps), so
        that the run allocates nothing — needed before such a run is captured into a hipGraph on a fresh context."""
        with torch.cuda.device(self.device):
            self._bind_stream()
            self._check(self.lib.rsik_control_continuous_reserve(self._h, int(n), int(n_steps)))

    def control_continuous_release(self) -> None:
        """rsik_control_continuous_release: waits for the device and frees the workspace(s) continuous runs keep in the context
        (hipGraphs captured from such runs must not be replayed afterwards)."""
        with torch.cuda.device(self.device):
            self._check(self.lib.rsik_control_continuous_release(self._h))

    # ------------------------------------------------------------------ checks
    def _dev_f64(self, t: torch.Tensor, shape: Sequence[int], name: str) -> torch.Tensor:
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.asarray(t, dtype=np.float64))
        t = t.to(device=self.device, dtype=_F64)
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t.contiguous()

    def _dev_cols(self, t: torch.Tensor, rows: int, n: int, name: str) -> torch.Tensor:
        """SoA input [rows, n]: the ABI takes one pointer per column array, so any view whose rows are unit-stride
        (e.g. a column slice columns[:, lo:hi] of a larger batch) is passed as it is, without a copy."""
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.asarray(t, dtype=np.float64))
        t = t.to(device=self.device, dtype=_F64)
        if tuple(t.shape) != (rows, n):
            raise ValueError(f"{name}: expected shape {(rows, n)}, got {tuple(t.shape)}")
        if n > 1 and t.stride(1) != 1:
            t = t.contiguous()
        return t

    def _dev_u8(self, t, n: int, name: str) -> torch.Tensor:
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.asarray(t, dtype=np.uint8))
        t = t.to(device=self.device, dtype=_U8)
        if tuple(t.shape) != (n,):
            raise ValueError(f"{name}: expected shape ({n},), got {tuple(t.shape)}")
        return t.contiguous()

    def _out_buf(self, out: Optional[Dict[str, torch.Tensor]], name: str, shape: Sequence[int], dtype: torch.dtype) -> torch.Tensor:
        """The caller's `out[name]` if given — it must be exactly what the kernel writes (this device, dtype, shape,
        contiguous: the kernels get a raw pointer) — else a fresh buffer."""
        t = None if out is None else out.get(name, None)
        shape = tuple(int(v) for v in shape)
        if t is None:
            return torch.empty(shape, dtype=dtype, device=self.device)
        if not isinstance(t, torch.Tensor):
            raise ValueError(f"out[{name!r}] must be a torch tensor")
        if t.device != self.device or t.dtype != dtype or tuple(t.shape) != shape or not t.is_contiguous():
            raise ValueError(f"out[{name!r}] must be a contiguous {dtype} tensor of shape {shape} on {self.device}; got "
                             f"{t.dtype} {tuple(t.shape)} on {t.device}{'' if t.is_contiguous() else ' (not contiguous)'}")
        return t

    # ------------------------------------------------------------------ rsik_solve
    def solve(
        self,
        pose_soa: torch.Tensor,
        arm: Optional[torch.Tensor] = None,
        arm_uniform: int = 0,
        theta_policy: int = _abi.THETA_INTERVAL0,
        theta_in: Optional[torch.Tensor] = None,
        previous_joints: Optional[Sequence[float]] = None,
        want_elbow: bool = True,
        out: Optional[Dict[str, torch.Tensor]] = None,
        plan_only: bool = False,
    ) -> Dict[str, torch.Tensor]:
        """pose_soa: [6, n] float64 (rows px,py,pz,roll,pitch,yaw).  Returns joints [n,7], interval [n,2],
        elbow [n,3], reachable [n] u8, state [n] u8 (device tensors, asynchronous on the current stream).
        plan_only=True launches nothing and adds res["launch"], a zero-overhead re-launch callable (see plan())."""
        if pose_soa.dim() != 2 or pose_soa.shape[0] != 6:
            raise ValueError("pose_soa must have shape [6, n]")
        n = int(pose_soa.shape[1])
        pose_soa = self._dev_cols(pose_soa, 6, n, "pose_soa")
        if arm is not None:
            arm = self._dev_u8(arm, n, "arm")
        if theta_policy in (_abi.THETA_EXPLICIT, _abi.THETA_FRACTION):
            if theta_in is None:
                raise ValueError("theta_in is required for this theta policy")
            theta_in = self._dev_f64(theta_in, (n,), "theta_in")
        else:
            theta_in = None
        none = theta_policy == _abi.THETA_NONE
        joints = None if none else self._out_buf(out, "joints", (n, 7), _F64)
        elbow = self._out_buf(out, "elbow", (n, 3), _F64) if (not none and want_elbow) else None
        interval = self._out_buf(out, "interval", (n, 2), _F64)
        reachable = self._out_buf(out, "reachable", (n,), _U8)
        state = self._out_buf(out, "state", (n,), _U8)
        cols = (C.c_void_p * 6)(*[pose_soa[k].data_ptr() for k in range(6)])
        prev = None
        if previous_joints is not None:
            pj = np.ascontiguousarray(previous_joints, dtype=np.float64)
            if pj.shape != (7,):
                raise ValueError("previous_joints must have 7 entries")
            prev = pj.ctypes.data_as(C.POINTER(C.c_double))
        cargs = (n, cols, _ptr(arm), int(arm_uniform), int(theta_policy), _ptr(theta_in), prev,
                 _ptr(joints), _ptr(interval), _ptr(elbow), _ptr(reachable), _ptr(state))
        res = {"interval": interval, "reachable": reachable, "state": state}
        if plan_only:
            res["launch"] = self.plan("rsik_solve", *cargs)
            res["_keepalive"] = (pose_soa, arm, theta_in, cols, prev)
        else:
            with torch.cuda.device(self.device):
                self._bind_stream()
                self._check(self.lib.rsik_solve(self._h, *cargs))
        if joints is not None:
            res["joints"] = joints
        if elbow is not None:
            res["elbow"] = elbow
        return res

    def plan(self, fn_name: str, *args):
        """Binds one C-ABI call with all its arguments once; the returned callable re-issues exactly that launch (a few
        microseconds of host time per call — the hot loop of a caller that re-solves resident buffers, e.g. bench.py).
        `launch()` enqueues on the stream that was current at PLANNING time, whatever other calls have done to the
        context's stream since; `launch(stream=handle)` enqueues on another hipStream_t (a capture stream when the
        launches are recorded into a hipGraph).  The plan is tied to the arm constants uploaded at planning time: it
        raises if set_arm() has changed them since.  Keep the tensors alive while the plan is used."""
        fn = getattr(self.lib, fn_name)
        set_stream = self.lib.rsik_set_stream
        with torch.cuda.device(self.device):
            planned = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        h, check, gen = self._h, self._check, self._arm_gen

        def launch(stream: Optional[int] = None) -> None:
            if self._arm_gen != gen:
                raise RuntimeError("HipSolver.plan: the arm constants were changed after this launch was planned")
            set_stream(h, planned if stream is None else C.c_void_p(stream))
            rc = fn(h, *args)
            if stream is not None:  # a caller's (capture) stream may not outlive the call: never leave the context on it
                set_stream(h, planned)
            if rc != _abi.RSIK_OK:
                check(rc)

        return launch

    # ------------------------------------------------------------------ rsik_control_discrete
    def control_discrete(
        self,
        m12_soa: torch.Tensor,
        arm: Optional[torch.Tensor] = None,
        arm_uniform: int = 0,
        nb_search_points: int = 20,
        preferred_theta: float = -4 * np.pi / 6,
        constrained_mode: int = _abi.MODE_UNCONSTRAINED,
        previous_sol: Optional[np.ndarray] = None,
        current_joints: Optional[torch.Tensor] = None,
        orbita3d_max_angle: float = float(np.deg2rad(42.5)),
        out: Optional[Dict[str, torch.Tensor]] = None,
        plan_only: bool = False,
    ) -> Dict[str, torch.Tensor]:
        """m12_soa: [12, n] float64 (R row-major, then translation)."""
        if m12_soa.dim() != 2 or m12_soa.shape[0] != 12:
            raise ValueError("m12_soa must have shape [12, n]")
        n = int(m12_soa.shape[1])
        m12_soa = self._dev_cols(m12_soa, 12, n, "m12_soa")
        if nb_search_points < 2:
            raise ValueError("nb_search_points must be >= 2")
        if arm is not None:
            arm = self._dev_u8(arm, n, "arm")
        if current_joints is not None:
            current_joints = self._dev_f64(current_joints, (n, 7), "current_joints")
        ps = np.ascontiguousarray(previous_sol, dtype=np.float64)
        if ps.shape != (2, 7):
            raise ValueError("previous_sol must have shape (2, 7)")
        joints = self._out_buf(out, "joints", (n, 7), _F64)
        reachable = self._out_buf(out, "reachable", (n,), _U8)
        state = self._out_buf(out, "state", (n,), _U8)
        emergency = self._out_buf(out, "emergency", (n,), _U8)
        cols = (C.c_void_p * 12)(*[m12_soa[k].data_ptr() for k in range(12)])
        cargs = (n, cols, _ptr(arm), int(arm_uniform), int(nb_search_points), float(preferred_theta), int(constrained_mode),
                 ps.ctypes.data_as(C.POINTER(C.c_double)), _ptr(current_joints), float(orbita3d_max_angle), _ptr(joints),
                 _ptr(reachable), _ptr(state), _ptr(emergency))
        res = {"joints": joints, "reachable": reachable, "state": state, "emergency": emergency}
        if plan_only:
            res["launch"] = self.plan("rsik_control_discrete", *cargs)
            res["_keepalive"] = (m12_soa, arm, current_joints, cols, ps)
        else:
            with torch.cuda.device(self.device):
                self._bind_stream()
                self._check(self.lib.rsik_control_discrete(self._h, *cargs))
        return res

    # ------------------------------------------------------------------ rsik_control_continuous_step
    def new_continuous_state(self, n: int) -> torch.Tensor:
        return torch.zeros((_abi.CONT_STATE_ROWS, n), dtype=_F64, device=self.device)

    def control_continuous_step(
        self,
        m12_soa: torch.Tensor,
        cont_state: torch.Tensor,
        preferred_theta_self: Sequence[float],
        arm: Optional[torch.Tensor] = None,
        arm_uniform: int = 0,
        timed_out: Optional[torch.Tensor] = None,
        preferred_theta: float = -4 * np.pi / 6,
        constrained_mode: int = _abi.MODE_UNCONSTRAINED,
        d_theta_max: float = 0.01,
        current_joints: Optional[torch.Tensor] = None,
        current_pose_m12: Optional[torch.Tensor] = None,
        orbita3d_max_angle: float = float(np.deg2rad(42.5)),
        out: Optional[Dict[str, torch.Tensor]] = None,
    ) -> Dict[str, torch.Tensor]:
        """One control step for n trajectories; `cont_state` ([RSIK_CONT_STATE_ROWS, n], see include/rsik.h) is updated in place."""
        if m12_soa.dim() != 2 or m12_soa.shape[0] != 12:
            raise ValueError("m12_soa must have shape [12, n]")
        n = int(m12_soa.shape[1])
        m12_soa = self._dev_f64(m12_soa, (12, n), "m12_soa")
        if (not isinstance(cont_state, torch.Tensor) or cont_state.dtype != _F64 or cont_state.device != self.device
                or tuple(cont_state.shape) != (_abi.CONT_STATE_ROWS, n) or not cont_state.is_contiguous()):
            raise ValueError(f"cont_state must be a contiguous float64 [{_abi.CONT_STATE_ROWS}, {n}] tensor on {self.device}")
        if arm is not None:
            arm = self._dev_u8(arm, n, "arm")
        if timed_out is not None:
            timed_out = self._dev_u8(timed_out, n, "timed_out")
        if current_joints is not None:
            current_joints = self._dev_f64(current_joints, (n, 7), "current_joints")
        cp = None
        if current_pose_m12 is not None:
            current_pose_m12 = self._dev_f64(current_pose_m12, (12, n), "current_pose_m12")
            cp = (C.c_void_p * 12)(*[current_pose_m12[k].data_ptr() for k in range(12)])
        pts = np.ascontiguousarray(preferred_theta_self, dtype=np.float64)
        if pts.shape != (2,):
            raise ValueError("preferred_theta_self must have 2 entries (r, l)")
        joints = self._out_buf(out, "joints", (n, 7), _F64)
        reachable = self._out_buf(out, "reachable", (n,), _U8)
        state = self._out_buf(out, "state", (n,), _U8)
        cols = (C.c_void_p * 12)(*[m12_soa[k].data_ptr() for k in range(12)])
        with torch.cuda.device(self.device):
            self._bind_stream()
            self._check(self.lib.rsik_control_continuous_step(
                self._h, n, cols, cp, _ptr(arm), int(arm_uniform), _ptr(timed_out), float(preferred_theta),
                pts.ctypes.data_as(C.POINTER(C.c_double)), int(constrained_mode), float(d_theta_max), _ptr(current_joints),
                float(orbita3d_max_angle), _ptr(cont_state), _ptr(joints), _ptr(reachable), _ptr(state)))
        return {"joints": joints, "reachable": reachable, "state": state}

    def control_continuous_run(
        self,
        m12_steps: torch.Tensor,
        cont_state: torch.Tensor,
        preferred_theta_self: Sequence[float],
        arm: Optional[torch.Tensor] = None,
        arm_uniform: int = 0,
        first_step_timed_out: bool = True,
        preferred_theta: float = -4 * np.pi / 6,
        constrained_mode: int = _abi.MODE_UNCONSTRAINED,
        d_theta_max: float = 0.01,
        current_joints: Optional[torch.Tensor] = None,
        current_pose_m12: Optional[torch.Tensor] = None,
        orbita3d_max_angle: float = float(np.deg2rad(42.5)),
        out: Optional[Dict[str, torch.Tensor]] = None,
        goals_resident: Optional[bool] = None,
    ) -> "ContinuousRunResult":
        """m12_steps: [n_steps, 12, n] float64 on the device.  All steps of all trajectories from one C call (the phased
        trajectory pipeline of rsik_control_continuous_run, or a launch of the step kernel per control step under
        RSIK_CONT_RUN_STEPS); returns joints [n_steps, n, 7], reachable / state [n_steps, n]; `cont_state` is updated in place.
        The result's attribute `run_form` says how the run was issued (rsik_control_continuous_last_form: _abi.CONT_FORM_*) — in particular
        when the library fell back to a launch per step because the arm's projection margin lets is_reachable_no_limits fail.
        `goals_resident` (None: RSIK_OPT_CONT_GOALS_RESIDENT as set on the context): the caller's promise that lets this run's
        prepare phase start beside the previous run's tail — include/rsik.h."""
        if m12_steps.dim() != 3 or m12_steps.shape[1] != 12:
            raise ValueError("m12_steps must have shape [n_steps, 12, n]")
        n_steps, _, n = (int(v) for v in m12_steps.shape)
        if m12_steps.dtype != _F64 or m12_steps.device != self.device or not m12_steps.is_contiguous():
            m12_steps = m12_steps.to(device=self.device, dtype=_F64).contiguous()
        if (cont_state.dtype != _F64 or cont_state.device != self.device or tuple(cont_state.shape) != (_abi.CONT_STATE_ROWS, n)
                or not cont_state.is_contiguous()):
            raise ValueError(f"cont_state must be a contiguous float64 [{_abi.CONT_STATE_ROWS}, {n}] tensor on {self.device}")
        if arm is not None:
            arm = self._dev_u8(arm, n, "arm")
        if current_joints is not None:
            current_joints = self._dev_f64(current_joints, (n, 7), "current_joints")
        cp = None
        if current_pose_m12 is not None:
            current_pose_m12 = self._dev_f64(current_pose_m12, (12, n), "current_pose_m12")
            cp = (C.c_void_p * 12)(*[current_pose_m12[k].data_ptr() for k in range(12)])
        pts = np.ascontiguousarray(preferred_theta_self, dtype=np.float64)
        joints = self._out_buf(out, "joints", (n_steps, n, 7), _F64)
        reachable = self._out_buf(out, "reachable", (n_steps, n), _U8)
        state = self._out_buf(out, "state", (n_steps, n), _U8)
        before = None
        if goals_resident is not None:
            before = self.get_option(_abi.OPT_CONT_GOALS_RESIDENT)
            self.set_option(_abi.OPT_CONT_GOALS_RESIDENT, int(bool(goals_resident)))
        try:
            with torch.cuda.device(self.device):
                self._bind_stream()
                self._check(self.lib.rsik_control_continuous_run(
                    self._h, n, n_steps, _ptr(m12_steps), cp, _ptr(arm), int(arm_uniform), int(bool(first_step_timed_out)),
                    float(preferred_theta), pts.ctypes.data_as(C.POINTER(C.c_double)), int(constrained_mode), float(d_theta_max),
                    _ptr(current_joints), float(orbita3d_max_angle), _ptr(cont_state), _ptr(joints), _ptr(reachable), _ptr(state)))
        finally:
            if before is not None:
                self.set_option(_abi.OPT_CONT_GOALS_RESIDENT, before)
        res = ContinuousRunResult({"joints": joints, "reachable": reachable, "state": state})
        res.run_form = self.continuous_last_form()
        return res

    def continuous_last_form(self) -> int:
        """How the last control_continuous_run of this context was issued: _abi.CONT_FORM_* (names: _abi.CONT_FORM_NAMES)."""
        return int(self.lib.rsik_control_continuous_last_form(self._h))

    # ------------------------------------------------------------------ solver-state entry points
    def new_solver_state(self, n: int) -> torch.Tensor:
        return torch.zeros((n, _abi.SOLVER_STATE_STRIDE), dtype=_F64, device=self.device)

    def reach_state(self, pose_soa: torch.Tensor, solver_state: torch.Tensor, arm: Optional[torch.Tensor] = None,
                    arm_uniform: int = 0, no_limits: bool = False) -> Dict[str, torch.Tensor]:
        n = int(pose_soa.shape[1])
        pose_soa = self._dev_f64(pose_soa, (6, n), "pose_soa")
        self._check_state(solver_state, n)
        if arm is not None:
            arm = self._dev_u8(arm, n, "arm")
        interval = torch.empty((n, 2), dtype=_F64, device=self.device)
        reachable = torch.empty((n,), dtype=_U8, device=self.device)
        state = torch.empty((n,), dtype=_U8, device=self.device)
        cols = (C.c_void_p * 6)(*[pose_soa[k].data_ptr() for k in range(6)])
        with torch.cuda.device(self.device):
            self._bind_stream()
            self._check(self.lib.rsik_reach_state(self._h, n, cols, _ptr(arm), int(arm_uniform), int(bool(no_limits)),
                                                  _ptr(solver_state), _ptr(interval), _ptr(reachable), _ptr(state)))
        return {"interval": interval, "reachable": reachable, "state": state}

    def joints_from_state(self, solver_state: torch.Tensor, theta: torch.Tensor, arm: Optional[torch.Tensor] = None,
                          arm_uniform: int = 0, previous_joints: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        n = int(solver_state.shape[0])
        self._check_state(solver_state, n)
        theta = self._dev_f64(theta, (n,), "theta")
        if arm is not None:
            arm = self._dev_u8(arm, n, "arm")
        if previous_joints is not None:
            previous_joints = self._dev_f64(previous_joints, (n, 7), "previous_joints")
        joints = torch.empty((n, 7), dtype=_F64, device=self.device)
        elbow = torch.empty((n, 3), dtype=_F64, device=self.device)
        with torch.cuda.device(self.device):
            self._bind_stream()
            self._check(self.lib.rsik_joints_from_state(self._h, n, _ptr(solver_state), _ptr(arm), int(arm_uniform),
                                                        _ptr(theta), _ptr(previous_joints), _ptr(joints), _ptr(elbow)))
        return {"joints": joints, "elbow": elbow}

    def elbow_from_state(self, solver_state: torch.Tensor, theta: torch.Tensor) -> torch.Tensor:
        n = int(solver_state.shape[0])
        self._check_state(solver_state, n)
        theta = self._dev_f64(theta, (n,), "theta")
        elbow = torch.empty((n, 3), dtype=_F64, device=self.device)
        with torch.cuda.device(self.device):
            self._bind_stream()
            self._check(self.lib.rsik_elbow_from_state(self._h, n, _ptr(solver_state), _ptr(theta), _ptr(elbow)))
        return elbow

    def stage(self, op: int, rows: torch.Tensor, arm_uniform: int = 0) -> torch.Tensor:
        """rsik_stage: one stage of SymbolicIK.is_reachable (include/rsik.h RSIK_STAGE_*) on explicit operands, row by row.
        rows: [n, doubles the stage reads] float64 on the device (or pinned host memory); returns [n, doubles it writes]."""
        need_in, need_out = _abi.STAGE_ROW[int(op)]
        if (not isinstance(rows, torch.Tensor) or rows.dtype != _F64 or rows.dim() != 2 or rows.shape[1] != need_in or not rows.is_contiguous()
                or not (rows.device == self.device or (rows.device.type == "cpu" and rows.is_pinned()))):
            raise ValueError(f"stage {op} takes a contiguous float64 [n, {need_in}] tensor on {self.device} (or in pinned host memory)")
        n = int(rows.shape[0])
        out = torch.empty((n, need_out), dtype=_F64, device=self.device)
        with torch.cuda.device(self.device):
            self._bind_stream()
            self._check(self.lib.rsik_stage(self._h, int(op), n, int(arm_uniform), _ptr(rows), need_in, _ptr(out), need_out))
        return out

    def _check_state(self, solver_state: torch.Tensor, n: int) -> None:
        if (not isinstance(solver_state, torch.Tensor) or solver_state.dtype != _F64 or solver_state.device != self.device
                or tuple(solver_state.shape) != (n, _abi.SOLVER_STATE_STRIDE) or not solver_state.is_contiguous()):
            raise ValueError(f"solver_state must be a contiguous float64 [{n}, {_abi.SOLVER_STATE_STRIDE}] tensor on {self.device}")

    # ------------------------------------------------------------------ goal matrix <-> Euler pose
    def set_option(self, option: int, value: int) -> None:
        """rsik_set_option, e.g. (_abi.OPT_EULER_ROUNDTRIP, 1): the control kernels run goal matrices through the
        reference's matrix -> Euler -> matrix round trip instead of consuming M[:3,:3] directly."""
        self._check(self.lib.rsik_set_option(self._h, int(option), int(value)))

    def get_option(self, option: int) -> int:
        v = C.c_int(0)
        self._check(self.lib.rsik_get_option(self._h, int(option), C.byref(v)))
        return int(v.value)

    def matrix_to_pose(self, m12_soa: torch.Tensor, identity_shortcut: bool = False) -> torch.Tensor:
        """Goal matrices [12,n] (R row-major, t) -> poses [6,n] (px,py,pz,roll,pitch,yaw): a batched
        utils.get_euler_from_homogeneous_matrix (utils.py:84-90); `identity_shortcut` adds control_ik.py:212-214."""
        if m12_soa.dim() != 2 or m12_soa.shape[0] != 12:
            raise ValueError("m12_soa must have shape [12, n]")
        n = int(m12_soa.shape[1])
        m12_soa = self._dev_f64(m12_soa, (12, n), "m12_soa")
        out = torch.empty((6, n), dtype=_F64, device=self.device)
        cin = (C.c_void_p * 12)(*[m12_soa[k].data_ptr() for k in range(12)])
        cout = (C.c_void_p * 6)(*[out[k].data_ptr() for k in range(6)])
        with torch.cuda.device(self.device):
            self._bind_stream()
            self._check(self.lib.rsik_matrix_to_pose(self._h, n, cin, 1 if identity_shortcut else 0, cout))
        return out

    # ------------------------------------------------------------------ forward kinematics
    def forward_kinematics(self, joints: torch.Tensor, arm: Optional[torch.Tensor] = None, arm_uniform: int = 0):
        """joints [n,7] -> (goal position [n,3], goal rotation [n,3,3]) in the torso frame (rsik_forward_kinematics)."""
        n = int(joints.shape[0])
        joints = self._dev_f64(joints, (n, 7), "joints")
        if arm is not None:
            arm = self._dev_u8(arm, n, "arm")
        pos = torch.empty((n, 3), dtype=_F64, device=self.device)
        rot = torch.empty((n, 3, 3), dtype=_F64, device=self.device)
        with torch.cuda.device(self.device):
            self._bind_stream()
            self._check(self.lib.rsik_forward_kinematics(self._h, n, _ptr(joints), _ptr(arm), int(arm_uniform), _ptr(pos), _ptr(rot)))
        return pos, rot

    def fk_residual(self, goal_soa: torch.Tensor, joints: torch.Tensor, arm: Optional[torch.Tensor] = None,
                    arm_uniform: int = 0) -> torch.Tensor:
        """FK(joints) against the goals they were solved for: err [n,2] = (position error m, rotation error rad).
        goal_soa: [6,n] poses (px,py,pz,roll,pitch,yaw) or [12,n] matrices (R row-major, t)."""
        if goal_soa.dim() != 2 or goal_soa.shape[0] not in (6, 12):
            raise ValueError("goal_soa must have shape [6, n] or [12, n]")
        rows, n = int(goal_soa.shape[0]), int(goal_soa.shape[1])
        goal_soa = self._dev_f64(goal_soa, (rows, n), "goal_soa")
        joints = self._dev_f64(joints, (n, 7), "joints")
        if arm is not None:
            arm = self._dev_u8(arm, n, "arm")
        err = torch.empty((n, 2), dtype=_F64, device=self.device)
        cols = (C.c_void_p * rows)(*[goal_soa[k].data_ptr() for k in range(rows)])
        kind = _abi.GOAL_M12 if rows == 12 else _abi.GOAL_POSE6
        with torch.cuda.device(self.device):
            self._bind_stream()
            self._check(self.lib.rsik_fk_residual(self._h, n, kind, cols, _ptr(joints), _ptr(arm), int(arm_uniform), _ptr(err)))
        return err

    # ------------------------------------------------------------------ measurement hooks
    def clock_monitor(self, seconds: float, n_waves: int = 64, stream: Optional[torch.cuda.Stream] = None):
        """Starts rsik_debug_math op 8 on `stream` (a side stream, so that it runs BESIDE whatever the main stream is
        doing): n_waves one-wave workgroups each wait `seconds` (clamped to 5 s) and report shader-clock and 100 MHz
        ticks.  Returns (core_ticks, real_ticks) device tensors, valid once `stream` has been synchronised;
        core clock in GHz = core_ticks / real_ticks * 0.1."""
        ticks = torch.tensor([float(seconds) * 1e8], dtype=_F64, device=self.device)
        o0 = torch.zeros((n_waves,), dtype=_F64, device=self.device)
        o1 = torch.zeros((n_waves,), dtype=_F64, device=self.device)
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        st.wait_stream(torch.cuda.current_stream(self.device))  # the tick count must have been written
        with torch.cuda.device(self.device):
            self._check(self.lib.rsik_set_stream(self._h, C.c_void_p(st.cuda_stream)))
            self._check(self.lib.rsik_debug_math(self._h, 8, int(n_waves), _ptr(ticks), None, _ptr(o0), _ptr(o1)))
            self._bind_stream()
        for t in (ticks, o0, o1):
            t.record_stream(st)
        return o0, o1

    def build_id(self) -> str:
        return (self.lib.rsik_build_id() or b"").decode()

    # ------------------------------------------------------------------ test hook
    def debug_math(self, op: int, a: torch.Tensor, b: Optional[torch.Tensor] = None):
        """Evaluates the kernels' own elementary functions (csrc/rsik_math.hpp) on device arrays (rsik_debug_math)."""
        n = int(a.numel())
        a = self._dev_f64(a, (n,), "a")
        if b is not None:
            b = self._dev_f64(b, (n,), "b")
        o0 = torch.empty((n,), dtype=_F64, device=self.device)
        o1 = torch.empty((n,), dtype=_F64, device=self.device)
        with torch.cuda.device(self.device):
            self._bind_stream()
            self._check(self.lib.rsik_debug_math(self._h, int(op), n, _ptr(a), _ptr(b), _ptr(o0), _ptr(o1)))
        return o0, o1
