"""ctypes binding of the C ABI declared in include/rsik.h (csrc/librsik_hip.so).

There is no CPU fallback: if the HIP library is missing or no GPU is present the product
path raises.  (`load()` itself works on a GPU-less machine so the symbol table can be checked.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
LIB_PATH = os.path.join(_HERE, "csrc", "librsik_hip.so")


def use_library(path: str) -> None:
    """Loads an alternative build of the same ABI instead of the in-tree library (A/B timing of kernel variants,
    diagnostic probe builds: `bench.py --lib`, scripts/*).  Must be called before the first load()."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("use_library() must be called before the HIP library is loaded")
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    LIB_PATH = os.path.abspath(path)


ABI_VERSION = 7
RSIK_OK = 0
RSIK_E_INVALID, RSIK_E_NO_DEVICE, RSIK_E_HIP, RSIK_E_NOT_SET = -1, -2, -3, -4

THETA_INTERVAL0, THETA_EXPLICIT, THETA_FRACTION, THETA_NONE = 0, 1, 2, 3
MODE_UNCONSTRAINED, MODE_LOW_ELBOW = 0, 1
MODES = {"unconstrained": MODE_UNCONSTRAINED, "low_elbow": MODE_LOW_ELBOW}

_vp = C.c_void_p
_dp = C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol include/rsik.h declares
PROTOTYPES = {
    "rsik_abi_version": (C.c_int, []),
    "rsik_build_id": (C.c_char_p, []),
    "rsik_arm_consts_count": (C.c_int, []),
    "rsik_device_count": (C.c_int, []),
    "rsik_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "rsik_destroy": (C.c_int, [_vp]),
    "rsik_last_error": (C.c_char_p, [_vp]),
    "rsik_set_stream": (C.c_int, [_vp, _vp]),
    "rsik_sync": (C.c_int, [_vp]),
    "rsik_set_arm": (C.c_int, [_vp, C.c_int, _dp, C.c_int]),
    "rsik_set_option": (C.c_int, [_vp, C.c_int, C.c_int]),
    "rsik_get_option": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int)]),
    "rsik_matrix_to_pose": (C.c_int, [_vp, C.c_int64, C.POINTER(_vp), C.c_int, C.POINTER(_vp)]),
    "rsik_malloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "rsik_free": (C.c_int, [_vp, _vp]),
    "rsik_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "rsik_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "rsik_solve": (C.c_int, [_vp, C.c_int64, C.POINTER(_vp), _vp, C.c_int, C.c_int, _vp, _dp, _vp, _vp, _vp, _vp, _vp]),
    "rsik_control_discrete": (C.c_int, [_vp, C.c_int64, C.POINTER(_vp), _vp, C.c_int, C.c_int, C.c_double, C.c_int, _dp,
                                        _vp, C.c_double, _vp, _vp, _vp, _vp]),
    "rsik_control_continuous_step": (C.c_int, [_vp, C.c_int64, C.POINTER(_vp), C.POINTER(_vp), _vp, C.c_int, _vp, C.c_double, _dp,
                                               C.c_int, C.c_double, _vp, C.c_double, _vp, _vp, _vp, _vp]),
    "rsik_control_continuous_run": (C.c_int, [_vp, C.c_int64, C.c_int64, _vp, C.POINTER(_vp), _vp, C.c_int, C.c_int, C.c_double, _dp,
                                              C.c_int, C.c_double, _vp, C.c_double, _vp, _vp, _vp, _vp]),
    "rsik_control_continuous_last_form": (C.c_int, [_vp]),
    "rsik_control_continuous_reserve": (C.c_int, [_vp, C.c_int64, C.c_int64]),
    "rsik_control_continuous_release": (C.c_int, [_vp]),
    "rsik_stage": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_int, _vp, C.c_int, _vp, C.c_int]),
    "rsik_reach_state": (C.c_int, [_vp, C.c_int64, C.POINTER(_vp), _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "rsik_joints_from_state": (C.c_int, [_vp, C.c_int64, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "rsik_elbow_from_state": (C.c_int, [_vp, C.c_int64, _vp, _vp, _vp]),
    "rsik_forward_kinematics": (C.c_int, [_vp, C.c_int64, _vp, _vp, C.c_int, _vp, _vp]),
    "rsik_fk_residual": (C.c_int, [_vp, C.c_int64, C.c_int, C.POINTER(_vp), _vp, _vp, C.c_int, _vp]),
    "rsik_debug_math": (C.c_int, [_vp, C.c_int, C.c_int64, _vp, _vp, _vp, _vp]),
    "rsik_comm_unique_id": (C.c_int, [_vp]),
    "rsik_comm_init_rank": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    "rsik_comm_destroy": (C.c_int, [_vp, _vp]),
    "rsik_allgather": (C.c_int, [_vp, _vp, _vp, _vp, C.c_size_t]),
}

GOAL_POSE6, GOAL_M12 = 0, 1
(STAGE_POSE_IN_REACH, STAGE_WRIST_POSITION, STAGE_LIMITATION_CIRCLE, STAGE_INTERSECTION_CIRCLE, STAGE_CIRCLES_LINKED, STAGE_NEAREST_APPROACH,
 STAGE_CIRCLE_LINE, STAGE_ROTATION_FROM_VECTOR, STAGE_ANGLE_DIFF, STAGE_IS_VALID_ANGLE, STAGE_LIMIT_THETA_TO_INTERVAL, STAGE_IS_ELBOW_OK,
 STAGE_ALLOW_MULTITURN, STAGE_LIMIT_ORBITA3D_JOINTS, STAGE_MULTITURN_SAFETY_CHECK, STAGE_CONTINUITY_CHECK, STAGE_BEST_DISCRETE_THETA) = range(17)
STAGE_ROW = {0: (6, 5), 1: (6, 3), 2: (6, 7), 3: (3, 8), 4: (17, 3), 5: (12, 7), 6: (10, 7), 7: (3, 9),  # doubles in / out per row
             8: (2, 1), 9: (3, 1), 10: (4, 2), 11: (9, 1), 12: (14, 7), 13: (4, 3), 14: (10, 8), 15: (21, 8), 16: (18, 3)}
STAGE_IN_MAX, STAGE_OUT_MAX = 21, 9
OPT_EULER_ROUNDTRIP, OPT_SWEEP_MODE, OPT_NO_TIPZ, OPT_NO_MIRROR, OPT_CONT_RUN_MODE = 0, 1, 2, 3, 4
OPT_CONT_BLOCK_STEPS, OPT_CONT_PHASED_VARIANT, OPT_CONT_GOALS_RESIDENT = 5, 6, 7
(CONT_FORM_NONE, CONT_FORM_PHASED, CONT_FORM_PHASED_OVERLAPPED, CONT_FORM_PHASED_CAPTURED, CONT_FORM_STEPS,
 CONT_FORM_STEPS_NO_LIMITS_CAN_FAIL) = range(6)
CONT_FORM_NAMES = {0: "none", 1: "phased", 2: "phased, overlapping the run before", 3: "phased, captured", 4: "steps",
                   5: "steps (the arm's projection margin lets is_reachable_no_limits fail)"}
PHASED_EDGES_BY_EVENT, PHASED_NO_THETA_FIRST = 1, 2
CONT_RUN_AUTO, CONT_RUN_PHASED, CONT_RUN_STEPS = 0, 1, 2
EMERGENCY_SHOULDER_PITCH, EMERGENCY_ELBOW_YAW, EMERGENCY_WRIST_YAW, EMERGENCY_CONTINUITY = 1, 2, 4, 8
EULER_AUTO, EULER_ALWAYS, EULER_NEVER = 0, 1, 2

STATE_EMERGENCY, STATE_NOT_REACHABLE_NO_LIMITS, STATE_INVALID_INPUT = 8, 9, 10
SOLVER_STATE_STRIDE = 32
CONT_STATE_ROWS = 19


class RsikError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"rsik error {code}: {message}")
        self.code = code


def load() -> C.CDLL:
    """Loads the HIP library; raises if it has not been built (no fallback).  The in-tree library must have been
    built from the sources next to it (embedded source hash, build.py): a stale one is rebuilt when hipcc is there,
    and refused otherwise, so that tests and benchmarks never validate or time an old binary."""
    global _lib
    if _lib is None:
        in_tree = LIB_PATH == os.path.join(_HERE, "csrc", "librsik_hip.so")
        if in_tree:
            from . import build as _build

            if _build.needs_build():
                try:
                    _build.build()
                except Exception as e:  # no hipcc, or the compile failed
                    state = "is missing" if not os.path.exists(LIB_PATH) else "was built from other sources than the ones next to it"
                    raise ImportError(
                        f"{LIB_PATH} {state} and could not be rebuilt ({e}): build it with "
                        "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950).  "
                        "reachy2_symbolic_ik_amd has no CPU fallback.") from e
        elif not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if lib.rsik_abi_version() != ABI_VERSION:
            raise ImportError("librsik_hip.so ABI version mismatch")
        _lib = lib
    return _lib
