"""Multi-GPU: poses are independent, so the batch is block-sharded across ranks (one process per GPU) with no
data-path collective; the only exchange is the all-gather of the final arrays (RCCL over xGMI on MI355X —
torch.distributed backend "nccl"; "gloo" in the CPU tests).  The reference has nothing distributed
(single-threaded Python, one pose per call); this is the SURVEY 8(e) design.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_size(n: int, world: int) -> int:
    return (n + world - 1) // world


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block split: rank r owns rows [r*S, min((r+1)*S, n)) with S = ceil(n / world).
    Only trailing ranks can be short or empty, so the concatenation of the S-padded shards, cut at n, is the
    full array — the all-gather can write straight into the [world*S, ...] result buffer."""
    s = shard_size(n, world)
    lo = min(rank * s, n)
    return lo, min(lo + s, n)


def all_gather_rows(local: torch.Tensor, n_total: int, group: Optional[dist.ProcessGroup] = None,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """All-gathers row-sharded `local` ([rows_of_this_rank, ...]) into the full [n_total, ...] tensor on every rank."""
    world = dist.get_world_size(group)
    s = shard_size(n_total, world)
    tail = tuple(local.shape[1:])
    if local.shape[0] != s:  # short / empty trailing shard: pad to the common shard size
        padded = local.new_zeros((s,) + tail)
        padded[: local.shape[0]] = local
        local = padded
    if out is None:
        out = local.new_empty((world * s,) + tail)
    elif out.shape[0] != world * s or tuple(out.shape[1:]) != tail:
        raise ValueError(f"out must have shape {(world * s,) + tail}")
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out[:n_total]


def solve_sharded(solve_fn: Callable[[torch.Tensor], Dict[str, torch.Tensor]], columns: torch.Tensor,
                  group: Optional[dist.ProcessGroup] = None,
                  gather: Iterable[str] = ("joints", "reachable", "state"), chunks: int = 1) -> Dict[str, torch.Tensor]:
    """columns: the full SoA input [C, n] (present on every rank).  Each rank solves its block with `solve_fn`
    (e.g. SymbolicIK.solve_batch) and the arrays named in `gather` are all-gathered; returns full-size arrays.

    chunks > 1: the block is solved in `chunks` pieces and the all-gather of piece k is issued
    asynchronously as soon as its solve has been queued, so it travels over xGMI while piece k+1 is being solved; every
    piece lands directly in its place of the full-size result (no staging copy)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = int(columns.shape[1])
    lo, hi = shard_range(n, rank, world)
    gather = tuple(gather)
    if chunks <= 1:
        local = solve_fn(columns[:, lo:hi].contiguous())
        return {k: all_gather_rows(local[k], n, group) for k in gather}
    s = shard_size(n, world)
    cs = (s + chunks - 1) // chunks          # rows per piece; every rank cuts its (padded) block at the same places
    full: Dict[str, torch.Tensor] = {}
    pending = []
    for c in range(chunks):
        a, b = c * cs, min((c + 1) * cs, s)  # piece rows inside a block
        if a >= b:
            break
        mine_lo, mine_hi = min(lo + a, hi), min(lo + b, hi)
        local = solve_fn(columns[:, mine_lo:mine_hi].contiguous())
        for k in gather:
            t = local[k]
            if k not in full:
                full[k] = t.new_zeros((world * s,) + tuple(t.shape[1:]))
            piece = t
            if t.shape[0] != b - a:          # short / empty trailing piece of a short trailing block: pad
                piece = t.new_zeros((b - a,) + tuple(t.shape[1:]))
                piece[: t.shape[0]] = t
            views = [full[k][r * s + a: r * s + b] for r in range(world)]
            pending.append(dist.all_gather(views, piece.contiguous(), group=group, async_op=True))
    for w in pending:
        w.wait()
    return {k: full[k][:n] for k in gather}
