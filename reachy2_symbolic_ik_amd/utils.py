"""Data-format helpers either side of the solve path, mirroring the reference's utils module for callers that build
goal matrices or read URDFs (reachy2_symbolic_ik/utils.py:12-23, 84-90, 661-694).  Host-side packing only: every
solve runs in the HIP kernels (there is no CPU implementation of the path in this package)."""
from __future__ import annotations

from typing import Tuple

import numpy as np
import numpy.typing as npt

from .constants import get_ik_parameters_from_urdf, parse_vector  # noqa: F401  (utils.py:661-694)


def make_homogenous_matrix_from_rotation_matrix(
    position: npt.NDArray[np.float64], rotation_matrix: npt.NDArray[np.float64]
) -> npt.NDArray[np.float64]:
    """4x4 homogeneous matrix from a 3x3 rotation and a position (utils.py:12-23)."""
    M = np.eye(4)
    M[:3, :3] = np.asarray(rotation_matrix, dtype=np.float64)[:3, :3]
    M[:3, 3] = np.asarray(position, dtype=np.float64)[:3]
    return M


_default_ik = None


def rotation_matrix_from_vector(vect: npt.NDArray[np.float64]) -> npt.NDArray[np.float64]:
    """The rotation that takes e_x to vect / |vect| (utils.py:59-81) — evaluated by the device's stage kernel (rsik_stage,
    RSIK_STAGE_ROTATION_FROM_VECTOR; the stage reads no arm constant) through a solver object this module creates on first use."""
    global _default_ik
    if _default_ik is None:
        import contextlib
        import io

        from . import _abi  # noqa: F401
        from .symbolic_ik import SymbolicIK

        with contextlib.redirect_stdout(io.StringIO()):
            _default_ik = SymbolicIK()
    from . import _abi

    return _default_ik._stage(_abi.STAGE_ROTATION_FROM_VECTOR, vect).reshape(3, 3)


def get_euler_from_homogeneous_matrix(
    homogeneous_matrix: npt.NDArray[np.float64], degrees: bool = False
) -> Tuple[npt.NDArray[np.float64], npt.NDArray[np.float64]]:
    """(position, extrinsic xyz Euler angles) of a 4x4 pose matrix (utils.py:84-90; scipy's conventions, including
    its gimbal-lock rule, since callers compare against scipy-produced angles)."""
    from scipy.spatial.transform import Rotation

    M = np.asarray(homogeneous_matrix, dtype=np.float64)
    return M[:3, 3], Rotation.from_matrix(M[:3, :3]).as_euler("xyz", degrees=degrees)
