"""MI355X-native batched analytic IK for the Reachy 2 arms.

Drop-in for the analytic solve path of pollen-robotics/reachy2_symbolic_ik
(SymbolicIK.is_reachable + theta_to_joints_func, ControlIK discrete mode), backed by
hand-written HIP kernels for gfx950 behind the C ABI of include/rsik.h.
"""
from .constants import STATE_STRINGS, ArmGeometry, default_ik_parameters  # noqa: F401

__all__ = ["SymbolicIK", "DualArmIK", "ControlIK", "HipSolver", "ArmGeometry", "STATE_STRINGS", "default_ik_parameters"]


def __getattr__(name):
    # torch / the HIP library are imported lazily so that host-only helpers stay importable
    if name == "SymbolicIK":
        from .symbolic_ik import SymbolicIK

        return SymbolicIK
    if name == "DualArmIK":
        from .symbolic_ik import DualArmIK

        return DualArmIK
    if name == "ControlIK":
        from .control_ik import ControlIK

        return ControlIK
    if name == "HipSolver":
        from .backend import HipSolver

        return HipSolver
    raise AttributeError(name)
