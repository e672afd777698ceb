"""Per-call latency of the scalar drop-in API (harness shaped like the reference's src/benchmark/ik_benchmarks.py:12-33)."""
import contextlib, io, sys, time
import numpy as np
sys.path.insert(0, ".")
from reachy2_symbolic_ik_amd import SymbolicIK, ControlIK
with contextlib.redirect_stdout(io.StringIO()):
    ik = SymbolicIK()
    c = ControlIK(urdf_path="config_files/reachy2_ik_minimal.urdf")
pose = np.array([[0.3, -0.1, 0.1], [np.radians(20), np.radians(-50), np.radians(20)]])
pose2 = np.array([[0.55, -0.3, -0.15], [0, -np.pi / 2, 0]])
def timeit(fn, batches=10, n=100):
    """median over batches of n calls (a process sees one ~75 ms one-off runtime stall in its first second)"""
    fn(); out = []
    for _ in range(batches):
        t = time.perf_counter()
        for _ in range(n): fn()
        out.append((time.perf_counter() - t) / n * 1e6)
    return sorted(out)[len(out) // 2]
print("is_reachable            %.1f us" % timeit(lambda: ik.is_reachable(pose2)))
ok, itv, fn, st = ik.is_reachable(pose2)
print("get_joints              %.1f us" % timeit(lambda: fn(itv[0])))
print("get_elbow_position      %.1f us" % timeit(lambda: ik.get_elbow_position(0.3)))
from scipy.spatial.transform import Rotation as R
M = np.eye(4); M[:3, :3] = R.from_euler("xyz", pose2[1]).as_matrix(); M[:3, 3] = pose2[0]
print("ControlIK discrete      %.1f us" % timeit(lambda: c.symbolic_inverse_kinematics("r_arm", M, "discrete")))
print("ControlIK continuous    %.1f us" % timeit(lambda: c.symbolic_inverse_kinematics("r_arm", M, "continuous")))
# the stage methods the reference's harness times one by one (src/benchmark/ik_benchmarks.py:36-130), chained as it chains them
from reachy2_symbolic_ik_amd.utils import rotation_matrix_from_vector
ik.wrist_position = ik.get_wrist_position(pose)
lc, ic = ik.get_limitation_wrist_circle(pose), ik.get_intersection_circle(pose)
q, v = ik.points_of_nearest_approach(lc[0], lc[2], ic[0], ic[2])
print("is_pose_in_robot_reach  %.1f us" % timeit(lambda: ik.is_pose_in_robot_reach(pose)))
print("get_wrist_position      %.1f us" % timeit(lambda: ik.get_wrist_position(pose)))
print("get_limitation_wrist_circle %.1f us" % timeit(lambda: ik.get_limitation_wrist_circle(pose)))
print("get_intersection_circle %.1f us" % timeit(lambda: ik.get_intersection_circle(pose)))
print("are_circles_linked      %.1f us" % timeit(lambda: ik.are_circles_linked(ic, lc)))
print("points_of_nearest_approach %.1f us" % timeit(lambda: ik.points_of_nearest_approach(lc[0], lc[2], ic[0], ic[2])))
print("intersection_circle_line_3d_vd %.1f us" % timeit(lambda: ik.intersection_circle_line_3d_vd(lc[0], lc[1], v, q)))
print("rotation_matrix_from_vector %.1f us" % timeit(lambda: rotation_matrix_from_vector(lc[2])))
