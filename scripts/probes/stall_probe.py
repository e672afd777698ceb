#!/usr/bin/env python3
"""The periodic hiccup of launch-by-launch issue: config-5 passes (17 launches each) issued back to back, the host time of every
run call; which calls block and for how long.  usage: stall_probe.py [passes=700]   (environment: HIP_FORCE_DEV_KERNARG etc. as given)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from reachy2_symbolic_ik_amd import ControlIK, _abi as A
n, N = 4096, 1000
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 700
traj = bench.make_config5_trajectories(n, N, device=0)
c = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
st0 = c.new_continuous_state("r_arm", n); st = st0.clone(); out = None
for _ in range(3):
    out = c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0], out=out)
torch.cuda.synchronize()
host = []
t_all = time.perf_counter()
for k in range(passes):
    t0 = time.perf_counter()
    st.copy_(st0)
    out = c.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0], out=out)
    host.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
total = (time.perf_counter() - t_all) * 1e3
slow = [(k, round(h, 2)) for k, h in enumerate(host) if h > 2.0]
print(f"HIP_FORCE_DEV_KERNARG={os.environ.get('HIP_FORCE_DEV_KERNARG')}: {passes} passes in {total:.1f} ms = {total / passes:.4f} ms per pass; "
      f"median host call {sorted(host)[len(host) // 2]:.3f} ms; calls over 2 ms: {slow}")
