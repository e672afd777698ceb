"""MI355X-native batched analytic IK for the Reachy 2 arms.

Drop-in for the analytic solve path of pollen-robotics/reachy2_symbolic_ik
(SymbolicIK.is_reachable + theta_to_joints_func, ControlIK discrete mode), backed by
hand-written HIP kernels for gfx950 behind the C ABI of include/rsik.h.
"""
import os as _os

# Kernel-argument buffers in device memory instead of host memory: the HIP runtime reads this when it initialises (the
# first HIP call of the process), so it is set at import, before torch touches the GPU.  The solve kernels take ~1 KB of
# arguments (both arm constant blocks); with host-resident kernarg every back-to-back launch costs 2-6 us more on MI355X
# (scripts/run_timeline.sh).  An explicit setting in the caller's environment wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

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
