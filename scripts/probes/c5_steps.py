import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from reachy2_symbolic_ik_amd import ControlIK, _abi
n, n_steps = 4096, 1000
traj = bench.make_config5_trajectories(n, n_steps, seed=20250204, device=0)
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
cont0 = ctrl.new_continuous_state("r_arm", n)
out = {"joints": torch.empty((n_steps, n, 7), dtype=torch.float64, device="cuda"),
       "reachable": torch.empty((n_steps, n), dtype=torch.uint8, device="cuda"),
       "state": torch.empty((n_steps, n), dtype=torch.uint8, device="cuda")}
cont = cont0.clone()
for mode, name in ((_abi.CONT_RUN_STEPS, "one step-kernel launch per control step"), (_abi.CONT_RUN_PHASED, "phased pipeline")):
    ctrl._solver.set_option(_abi.OPT_CONT_RUN_MODE, mode)
    def one():
        cont.copy_(cont0)
        ctrl.run_continuous_trajectories("r_arm", traj, cont, first_step_timed_out=True, current_pose=traj[0], out=out)
    for _ in range(3): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): one()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{name}: {ms:.3f} ms per 1000-step pass = {ms:.2f} us per control step of {n} trajectories")
