import os, sys, time, torch
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 1:
    from reachy2_symbolic_ik_amd import _abi
    _abi.use_library(os.path.abspath(sys.argv[1]))
import bench
from reachy2_symbolic_ik_amd import ControlIK
n, N = 4096, 1000
traj = bench.make_config5_trajectories(n, N, device=0)
c = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
st0 = c.new_continuous_state("r_arm", n)
st = st0.clone()
out = c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0])
torch.cuda.synchronize()
st_end = st.clone()
def one():
    st.copy_(st_end)
    c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=False, current_pose=traj[0], out=out)
for _ in range(3): one()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): one()
torch.cuda.synchronize()
name = sys.argv[1] if len(sys.argv) > 1 else "tree"
print(f"{name}: pass in which most trajectories latch at its first step {(time.perf_counter() - t0) / 10 * 1e3:.4f} ms, latched {int(st[9].sum())} of {n}, checksum {float(out['joints'].sum()):.9e} states {torch.bincount(out['state'].flatten().to(torch.int64), minlength=11).tolist()}")
# ... and the passes after that one: the trajectories are latched when the run begins (the robot has stopped; C:205-210 until "unfreeze")
st_latched = st.clone()
def again():
    st.copy_(st_latched)
    c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=False, current_pose=traj[0], out=out)
for _ in range(3): again()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): again()
torch.cuda.synchronize()
print(f"{name}: pass that BEGINS with {int(st_latched[9].sum())} of {n} trajectories latched {(time.perf_counter() - t0) / 10 * 1e3:.4f} ms, checksum {float(out['joints'].sum()):.9e}")
# ... and once more from where THAT pass ended (the rest have latched on the jump at its first step)
st_all = st.clone()
def third():
    st.copy_(st_all)
    c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=False, current_pose=traj[0], out=out)
for _ in range(3): third()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): third()
torch.cuda.synchronize()
print(f"{name}: pass that BEGINS with {int(st_all[9].sum())} of {n} trajectories latched {(time.perf_counter() - t0) / 10 * 1e3:.4f} ms, checksum {float(out['joints'].sum()):.9e}")
