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
# ... and with EVERY trajectory latched when the run begins (the flag forced)
st_forced = st_all.clone()
st_forced[9] = 1.0
def forced():
    st.copy_(st_forced)
    c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=False, current_pose=traj[0], out=out)
for _ in range(3): forced()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): forced()
torch.cuda.synchronize()
print(f"{name}: pass that BEGINS with all {n} trajectories latched {(time.perf_counter() - t0) / 10 * 1e3:.4f} ms; the survivors of the passes above: init flag set on "
      f"{int((st_all[8, st_all[9] == 0] != 0).sum())} of {int((st_all[9] == 0).sum())}, has_previous on {int((st_all[10, st_all[9] == 0] != 0).sum())}")
# which of the two makes the mixed pass slow: the survivors' own state, or their sharing waves with latched trajectories?
fresh = c.new_continuous_state("r_arm", n)
surv = st_all[9] == 0
st_mix = st_all.clone()
st_mix[:, surv] = fresh[:, surv]          # the survivors start like new trajectories (re-initialised at the first step)
def mixed():
    st.copy_(st_mix)
    c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=False, current_pose=traj[0], out=out)
for _ in range(3): mixed()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): mixed()
torch.cuda.synchronize()
print(f"{name}: pass that BEGINS with {int(st_mix[9].sum())} latched and the others as new trajectories {(time.perf_counter() - t0) / 10 * 1e3:.4f} ms; latched at its end {int(st[9].sum())}")
j = out["joints"][:, surv, :]
print("survivors in the last pass: largest step between consecutive control steps per joint", (j[1:] - j[:-1]).abs().amax(dim=(0, 1)).cpu().numpy().round(3),
      " |joint| max", j.abs().amax(dim=(0, 1)).cpu().numpy().round(2))
# the survivors by themselves, from the state they carry
idx = torch.nonzero(surv).flatten()
tr_s = traj[:, :, idx].contiguous()
st_s0 = st_all[:, idx].contiguous()
from reachy2_symbolic_ik_amd import _abi as A
for mode_name, mode in (("phased", A.CONT_RUN_PHASED), ("steps", A.CONT_RUN_STEPS)):
    c._solver.set_option(A.OPT_CONT_RUN_MODE, mode)
    s2 = st_s0.clone()
    for _ in range(2):
        s2.copy_(st_s0); o = c.run_continuous_trajectories("r_arm", tr_s, s2, first_step_timed_out=False, current_pose=tr_s[0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        s2.copy_(st_s0); o = c.run_continuous_trajectories("r_arm", tr_s, s2, first_step_timed_out=False, current_pose=tr_s[0])
    torch.cuda.synchronize()
    jj = o["joints"]
    dj = (jj[1:] - jj[:-1]).abs()
    print(f"survivors alone ({len(idx)}), {mode_name}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per pass; steps over 0.25 rad in joints 0-3: {int((dj[..., :4] > 0.25).sum())}, "
          f"NaN joints {int(jj.isnan().sum())}, states {torch.bincount(o['state'].flatten().to(torch.int64), minlength=11).tolist()}, first-step jump max {float((jj[0] - st_s0[1:8].T).abs().max()):.3f}")
c._solver.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_AUTO)
