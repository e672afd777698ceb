"""world_size-2 CPU (gloo) test of the sharding + all-gather path.  The per-rank solve is a stand-in that calls the
CPU checker (there is no CPU product path); what is under test is reachy2_symbolic_ik_amd/distributed.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from reachy2_symbolic_ik_amd.distributed import all_gather_rows, shard_range, shard_size, solve_sharded


def test_shard_ranges_cover_everything():
    for n in (0, 1, 2, 7, 8, 9, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            s = shard_size(n, world)
            assert all(hi - lo <= s for lo, hi in spans)
            assert all(lo == min(r * s, n) for r, (lo, hi) in enumerate(spans))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, seed, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc

        rng = np.random.default_rng(seed)
        pos = np.array([0.25, -0.2, -0.15]) + rng.uniform(-0.3, 0.3, size=(n, 3))
        eul = np.array([0, -np.pi / 2, 0]) + rng.uniform(-0.8, 0.8, size=(n, 3))
        arm = (rng.uniform(size=n) < 0.5).astype(np.uint8)
        pos[arm == 1, 1] *= -1.0
        cols = torch.as_tensor(np.ascontiguousarray(np.concatenate([pos.T, eul.T, arm[None].astype(np.float64)], axis=0)))
        ar, al = orc.Arm("r_arm", 0.03), orc.Arm("l_arm", 0.03)

        def solve_fn(c):
            c = c.numpy()
            if c.shape[1] == 0:
                return {"joints": torch.empty((0, 7), dtype=torch.float64), "reachable": torch.empty((0,), dtype=torch.uint8),
                        "state": torch.empty((0,), dtype=torch.uint8)}
            r = orc.solve_batch(ar, al, c[:3].T, c[3:6].T, arm_id=c[6].astype(np.uint8))
            return {k: torch.as_tensor(r[k]) for k in ("joints", "reachable", "state")}

        full = solve_sharded(solve_fn, cols)
        ref = orc.solve_batch(ar, al, pos, eul, arm_id=arm)
        for chunks in (2, 3, 5):  # pieces solved one after the other, each all-gathered asynchronously into place
            piecewise = solve_sharded(solve_fn, cols, chunks=chunks)
            for k in ("joints", "reachable", "state"):
                assert piecewise[k].shape == full[k].shape
                np.testing.assert_array_equal(np.nan_to_num(piecewise[k].numpy(), nan=-99.0), np.nan_to_num(full[k].numpy(), nan=-99.0))
        assert full["joints"].shape == (n, 7) and full["reachable"].shape == (n,)
        np.testing.assert_array_equal(full["reachable"].numpy(), ref["reachable"])
        np.testing.assert_array_equal(full["state"].numpy(), ref["state"])
        np.testing.assert_array_equal(np.nan_to_num(full["joints"].numpy(), nan=-99.0), np.nan_to_num(ref["joints"], nan=-99.0))
        # preallocated result buffer path
        lo, hi = shard_range(n, rank, world)
        buf = torch.empty((world * shard_size(n, world), 7), dtype=torch.float64)
        got = all_gather_rows(torch.as_tensor(ref["joints"][lo:hi]), n, out=buf)
        assert got.data_ptr() == buf.data_ptr()
        np.testing.assert_array_equal(np.nan_to_num(got.numpy(), nan=-99.0), np.nan_to_num(ref["joints"], nan=-99.0))
        open(os.path.join(result_dir, f"ok_{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [1, 2, 257, 1000])
def test_sharded_solve_world2_gloo(tmp_path, n):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n, 1234 + n, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok_{r}") for r in range(world))
