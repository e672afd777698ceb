#!/bin/bash
# config-5 passes of the phased pipeline, issued launch by launch, for several block sizes; VARIANT = RSIK_OPT_CONT_PHASED_VARIANT bits
# (1 edges by event, 2 no theta-first)
for blk in "$@"; do
  echo "== block steps $blk  (variant: ${VARIANT:-0})"
  timeout -k 10 200 python - "$blk" <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from reachy2_symbolic_ik_amd import ControlIK, _abi as A
blk = int(sys.argv[1])
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
hs = ctrl._solver
n, N = 4096, 1000
traj = bench.make_config5_trajectories(n, N, device=0)
cont0 = ctrl.new_continuous_state("r_arm", n)
hs.set_option(A.OPT_CONT_RUN_MODE, A.CONT_RUN_PHASED)
hs.set_option(A.OPT_CONT_BLOCK_STEPS, blk)
hs.set_option(A.OPT_CONT_PHASED_VARIANT, int(os.environ.get("VARIANT", "0")))
st = cont0.clone()
out = None
def one():
    global out
    st.copy_(cont0)
    out = ctrl.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0], out=out)
for _ in range(5): one()
torch.cuda.synchronize()
best = 1e9
issue = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20): one()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    issue = min(issue, (t1 - t0) / 20 * 1e3)
print(f"   host issue time {issue:.4f} ms per pass")
print(f"phased eager, {blk} steps per block: {best:.4f} ms per pass, {n * N / best / 1e6:.2f} G steps/s  checksum {float(out['joints'][-1].nan_to_num().sum()):.6f}")
PY
done
