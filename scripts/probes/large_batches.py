#!/usr/bin/env python3
"""Batches whose arrays reach past 2^31 and 2^32 bytes: the same poses tiled `tiles` times through ONE launch, every tile's rows
compared with the first tile's bit for bit (a wrong 32-bit offset anywhere shows as a tile that differs or as rows never written).

    python scripts/probes/large_batches.py [tiles of 1 Mi poses, default 80] [trajectories of the continuous run, default 1048576]

rsik_solve: joints [n, 7] f64 = 4.7 GB at 80 Mi poses; rsik_control_discrete: goal matrices [12, n] = 8 GB; the continuous run:
n trajectories x 96 steps (joints [96, n, 7] = 4.5 GB at 1 Mi trajectories; its blocks are sized so that the sequential phases' 2 GB
buffer windows hold them, cont_plan in rsik_lib.hip)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from reachy2_symbolic_ik_amd import ControlIK, SymbolicIK  # noqa: E402

tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 80
n_traj = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
dev = torch.device("cuda", 0)
base = 1 << 20
bad = 0


def same_tiles(name, t, rows_axis=0):
    """t: [tiles * base, ...] along rows_axis; every tile against tile 0 (bytes)."""
    global bad
    t = t.movedim(rows_axis, 0).contiguous()
    v = t.view(torch.uint8).reshape(tiles, -1)
    diff = [k for k in range(1, tiles) if not torch.equal(v[k], v[0])]
    print(f"  {name:10s} {tuple(t.shape)} {t.numel() * t.element_size() / 2**30:6.2f} GiB: "
          + ("every tile identical to the first" if not diff else f"TILES DIFFER: {diff[:8]}"), flush=True)
    bad += 1 if diff else 0


# --- rsik_solve: all outcomes (an unfiltered sample), theta = interval[0]
rng = np.random.default_rng(5)
pos = np.array([0.0, -0.2, 0.0]) + rng.uniform(-0.7, 0.7, (base, 3))
eul = rng.uniform(-np.pi, np.pi, (base, 3))
soa1 = torch.as_tensor(np.concatenate([pos.T, eul.T]), device=dev)  # [6, base]
soa = soa1.repeat(1, tiles).contiguous()
r = bench._quiet(SymbolicIK, arm="r_arm", device=0)
n = base * tiles
print(f"rsik_solve, {n} poses ({tiles} tiles of {base}):", flush=True)
res = r.solve_batch(soa)
torch.cuda.synchronize()
for k in ("joints", "interval", "elbow", "reachable", "state"):
    same_tiles(k, res[k])
small = r.solve_batch(soa1)
torch.cuda.synchronize()
ok = all(torch.equal(res[k][:base].contiguous().view(torch.uint8), small[k].contiguous().view(torch.uint8)) for k in small)
print("  first tile == the same poses solved as a batch of their own:", ok, flush=True)
bad += 0 if ok else 1
del res, soa

# --- rsik_control_discrete
ctrl = bench._quiet(ControlIK, urdf_path=bench.URDF, device=0)
from reachy2_symbolic_ik_amd.control_ik import matrices_to_m12_soa  # noqa: E402

m1 = matrices_to_m12_soa(bench.make_config3_matrices(base, seed=11, device=0), dev)  # [12, base]
m12 = m1.repeat(1, tiles).contiguous()
ctrl.nb_search_points = 64
print(f"rsik_control_discrete, {n} matrices:", flush=True)
res = ctrl.symbolic_inverse_kinematics_batch("r_arm", m12)
torch.cuda.synchronize()
for k in ("joints", "reachable", "state"):
    same_tiles(k, res[k])
small = ctrl.symbolic_inverse_kinematics_batch("r_arm", m1)
torch.cuda.synchronize()
ok = all(torch.equal(res[k][:base].contiguous().view(torch.uint8), small[k].contiguous().view(torch.uint8)) for k in ("joints", "reachable", "state"))
print("  first tile == the same matrices as a batch of their own:", ok, flush=True)
bad += 0 if ok else 1
del res, m12

# --- the continuous run: n_traj trajectories (tiles of 4096) x 96 steps
t_tiles, n_steps = n_traj // 4096, 96
traj1 = bench.make_config5_trajectories(4096, n_steps, seed=3, device=0)  # [steps, 12, 4096]
traj = traj1.repeat(1, 1, t_tiles).contiguous()
tiles = t_tiles
print(f"rsik_control_continuous_run, {4096 * t_tiles} trajectories x {n_steps} steps:", flush=True)
st = ctrl.new_continuous_state("r_arm", 4096 * t_tiles)
out = ctrl.run_continuous_trajectories("r_arm", traj, st, first_step_timed_out=True, current_pose=traj[0])
torch.cuda.synchronize()
same_tiles("joints", out["joints"], rows_axis=1)
same_tiles("state", out["state"], rows_axis=1)
same_tiles("reachable", out["reachable"], rows_axis=1)
same_tiles("cont_state", st, rows_axis=1)
st1 = ctrl.new_continuous_state("r_arm", 4096)
o1 = ctrl.run_continuous_trajectories("r_arm", traj1, st1, first_step_timed_out=True, current_pose=traj1[0])
torch.cuda.synchronize()
ok = (torch.equal(out["state"][:, :4096], o1["state"]) and torch.equal(out["reachable"][:, :4096], o1["reachable"])
      and torch.equal(st[0, :4096].contiguous().view(torch.uint8), st1[0].contiguous().view(torch.uint8)))
dj = float((out["joints"][:, :4096] - o1["joints"]).abs().nan_to_num(0.0).max())
print(f"  first tile against the same 4096 trajectories as a run of their own: flags / states / theta identical {ok}, joints {dj:.1e} "
      f"(run form {out.run_form_name})", flush=True)
bad += 0 if ok and dj <= 1e-9 else 1
print("TOTAL", "ok" if bad == 0 else f"{bad} checks failed")
sys.exit(1 if bad else 0)
