"""Forward kinematics of the Reachy 2 arm in NumPy — an oracle-free check of the IK kernels.

Chain (derived from the frames the solver inverts, symbolic_ik.py:728-848):
  torso -> shoulder:  translate s, rotate Ms = R(offset deg) Ry(pi/2)
  shoulder pitch j0 (about y), roll j1 (about z), translate u along x
  elbow yaw j2 (about x, sign reversed), pitch j3 (about y), translate f along x
  wrist roll j4 (about z), pitch j5 (about y, joint sign reversed), translate tip_z along x
  goal frame in the tip frame: z_goal = -x_tip, x_goal = (0, sin j6, cos j6)
"""
import numpy as np


def _rx(a):
    c, s = np.cos(a), np.sin(a)
    o, z = np.ones_like(a), np.zeros_like(a)
    return np.stack([np.stack([o, z, z], -1), np.stack([z, c, -s], -1), np.stack([z, s, c], -1)], -2)


def _ry(a):
    c, s = np.cos(a), np.sin(a)
    o, z = np.ones_like(a), np.zeros_like(a)
    return np.stack([np.stack([c, z, s], -1), np.stack([z, o, z], -1), np.stack([-s, z, c], -1)], -2)


def _rz(a):
    c, s = np.cos(a), np.sin(a)
    o, z = np.ones_like(a), np.zeros_like(a)
    return np.stack([np.stack([c, -s, z], -1), np.stack([s, c, z], -1), np.stack([z, z, o], -1)], -2)


def forward_kinematics(joints, shoulder_position, shoulder_offset_deg, upper_arm, forearm, tip_z):
    """joints [n,7] -> (goal position [n,3], goal rotation [n,3,3]) in the torso frame (tip offset (0,0,tip_z))."""
    j = np.asarray(joints, dtype=np.float64)
    off = np.radians(np.asarray(shoulder_offset_deg, dtype=np.float64))
    one = np.ones(1)
    Ms = (_rz(off[2] * one) @ _ry(off[1] * one) @ _rx(off[0] * one) @ _ry(np.pi / 2 * one))[0]
    Gt = _ry(j[:, 0]) @ _rz(j[:, 1])            # (Rz(-j1) Ry(-j0))^T
    Ht = _rx(-j[:, 2]) @ _ry(j[:, 3])           # (Ry(-j3) Rx(j2))^T
    Kt = _rz(j[:, 4]) @ _ry(j[:, 5])            # (Ry(wp) Rz(-wr))^T with wr = j4, wp = -j5
    ex = np.array([1.0, 0.0, 0.0])
    R_tip = Ms @ Gt @ Ht @ Kt                   # tip frame axes in the torso frame
    p = np.asarray(shoulder_position) + np.einsum("ij,nj->ni", Ms, np.einsum("nij,nj->ni", Gt, upper_arm * ex + np.einsum(
        "nij,nj->ni", Ht, forearm * ex + np.einsum("nij,j->ni", Kt, tip_z * ex))))
    s6, c6 = np.sin(j[:, 6]), np.cos(j[:, 6])
    z = np.zeros_like(s6)
    x_goal = np.stack([z, s6, c6], -1)
    z_goal = np.stack([-np.ones_like(s6), z, z], -1)
    y_goal = np.cross(z_goal, x_goal)
    B = np.stack([x_goal, y_goal, z_goal], -1)  # goal axes as columns, in the tip frame
    return p, R_tip @ B
