import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from reachy2_symbolic_ik_amd import ControlIK, _abi as A
n_traj, n_steps = int(sys.argv[1]), int(sys.argv[2])
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
hs = ctrl._solver
traj = bench.make_config5_trajectories(n_traj, n_steps, device=0)
cont0 = ctrl.new_continuous_state("r_arm", n_traj)
hs.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_FLAGS)
st = cont0.clone()
import time
try:
    for k in range(4):
        t0 = time.perf_counter()
        o = ctrl.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0])
        t1 = time.perf_counter()
        hs.synchronize()
        print(f"flags mode OK: issue {1e3 * (t1 - t0):.3f} ms, pass {1e3 * (time.perf_counter() - t0):.3f} ms")
except Exception as e:
    print("ERR", e)
    G = (n_traj + 63) // 64; B = (n_steps + 63) // 64
    print("G", G, "B", B, "kSyncArrays 128: pdone [128,", 128 + B * G, ") jdone [", 128 + B * G, ",", 128 + 2 * B * G, ") tprog from", 128 + 2 * B * G)
