"""Multi-GPU: poses are independent, so the batch is sharded across ranks (one process per GPU) with no data-path
collective; the only exchange is the all-gather of the final arrays (RCCL over xGMI on MI355X — torch.distributed
backend "nccl"; "gloo" in the CPU tests).  The reference has nothing distributed (single-threaded Python, one pose
per call); this is the SURVEY 8(e) design.

Partition (`ShardPlan`): the pose array is cut into `chunks` consecutive stripes of world * rows_per_piece rows and
rank r owns rows [r * rows_per_piece, (r + 1) * rows_per_piece) of every stripe (block-cyclic; chunks = 1 is the plain
contiguous block split of SURVEY 8(e)).  The all-gather of stripe c is then ONE all_gather_into_tensor whose output is
the contiguous stripe c of the full-size result in natural pose order, and whose input is this rank's own rows of that
stripe — the kernels write there directly, so there is no staging copy on either side — and it travels over xGMI
while the kernel solves stripe c + 1.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, Iterable, List, Optional, Tuple

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


@dataclass(frozen=True)
class ShardPlan:
    """Block-cyclic partition of n rows over `world` ranks in `chunks` stripes (see the module docstring)."""
    n: int
    world: int
    chunks: int = 1

    @property
    def rows_per_piece(self) -> int:
        return max(1, (self.n + self.world * self.chunks - 1) // (self.world * self.chunks))

    @property
    def stripe_rows(self) -> int:
        return self.world * self.rows_per_piece

    @property
    def padded_rows(self) -> int:
        """Rows of the full-size result buffer (rows >= n are padding, never read back)."""
        return self.chunks * self.stripe_rows

    def piece(self, rank: int, c: int) -> Tuple[int, int]:
        """Global rows [lo, lo + rows_per_piece) that `rank` owns in stripe c (may reach past n: padding)."""
        lo = c * self.stripe_rows + rank * self.rows_per_piece
        return lo, lo + self.rows_per_piece

    def owned(self, rank: int) -> List[Tuple[int, int]]:
        """The real (clipped to n) row ranges of `rank`, stripe by stripe."""
        out = []
        for c in range(self.chunks):
            lo, hi = self.piece(rank, c)
            out.append((min(lo, self.n), min(hi, self.n)))
        return out


def all_gather_rows(local: torch.Tensor, n_total: int, group: Optional[dist.ProcessGroup] = None,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """All-gathers row-sharded `local` ([rows_of_this_rank, ...], contiguous block split) into the full [n_total, ...]
    tensor on every rank."""
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


class ShardedBuffers:
    """Full-size result arrays of a sharded solve, [plan.padded_rows, ...] each, in natural pose order.
    `piece_views(c)` are this rank's rows of stripe c (what the kernels write), `stripe_views(c)` the whole stripe
    (what the all-gather of stripe c fills)."""

    def __init__(self, plan: ShardPlan, rank: int, spec: Dict[str, Tuple[Tuple[int, ...], torch.dtype]], device) -> None:
        self.plan, self.rank = plan, rank
        self.full = {k: torch.zeros((plan.padded_rows,) + tuple(tail), dtype=dt, device=device) for k, (tail, dt) in spec.items()}

    def piece_views(self, c: int) -> Dict[str, torch.Tensor]:
        lo, hi = self.plan.piece(self.rank, c)
        return {k: t[lo:hi] for k, t in self.full.items()}

    def stripe_views(self, c: int) -> Dict[str, torch.Tensor]:
        lo = c * self.plan.stripe_rows
        return {k: t[lo: lo + self.plan.stripe_rows] for k, t in self.full.items()}

    def result(self) -> Dict[str, torch.Tensor]:
        return {k: t[: self.plan.n] for k, t in self.full.items()}


def gather_stripe(buffers: ShardedBuffers, c: int, names: Iterable[str], group: Optional[dist.ProcessGroup] = None,
                  async_op: bool = True) -> list:
    """Issues the all-gather of stripe c for the named arrays: one all_gather_into_tensor per array, IN PLACE (the
    input is this rank's slice of the output, which is NCCL's / RCCL's in-place all-gather: no copy of the local
    rows).  Returns the work handles (empty when async_op is False)."""
    works = []
    pieces, stripes = buffers.piece_views(c), buffers.stripe_views(c)
    in_place = dist.get_backend(group) == "nccl"
    for k in names:
        src = pieces[k] if in_place else pieces[k].clone()  # gloo (CPU tests) does not promise in-place semantics
        w = dist.all_gather_into_tensor(stripes[k], src, group=group, async_op=async_op)
        if async_op:
            works.append(w)
    return works


def solve_sharded(solve_fn: Callable[[torch.Tensor, Dict[str, torch.Tensor]], None], columns: torch.Tensor,
                  spec: Optional[Dict[str, Tuple[Tuple[int, ...], torch.dtype]]] = None,
                  group: Optional[dist.ProcessGroup] = None, gather: Iterable[str] = ("joints", "state"),
                  chunks: int = 1) -> Dict[str, torch.Tensor]:
    """columns: the full SoA input [C, n] (present on every rank).  Every rank solves the rows it owns, stripe by
    stripe, with `solve_fn(columns[:, lo:hi], out)` — `out[k]` are [hi - lo, ...] views into the full-size result that
    the solve must fill (e.g. SymbolicIK.solve_batch(cols, out=out)) — and the arrays named in `gather` are
    all-gathered; returns the full-size arrays [n, ...] in pose order on every rank.

    chunks > 1: the all-gather of stripe c is issued asynchronously as soon as its solve has been queued,
    so it travels over xGMI while stripe c + 1 is being solved."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = int(columns.shape[1])
    plan = ShardPlan(n, world, max(1, int(chunks)))
    gather = tuple(gather)
    if spec is None:  # what rsik_solve / rsik_control_discrete can write per pose
        spec = {"joints": ((7,), torch.float64), "state": ((), torch.uint8), "reachable": ((), torch.uint8),
                "interval": ((2,), torch.float64), "elbow": ((3,), torch.float64)}
    unknown = [k for k in gather if k not in spec]
    if unknown:
        raise ValueError(f"solve_sharded: gather names {unknown} are not in spec (known: {sorted(spec)}); pass "
                         "spec={name: (trailing shape, dtype)} for other arrays")
    if not callable(solve_fn):
        raise TypeError("solve_sharded: solve_fn(columns, out) must be callable; it fills the views in `out` "
                        "(it does not return the arrays)")
    buffers = ShardedBuffers(plan, rank, {k: spec[k] for k in gather}, columns.device)
    pending = []
    for c in range(plan.chunks):
        lo, hi = plan.owned(rank)[c]
        if hi > lo:
            views = {k: v[: hi - lo] for k, v in buffers.piece_views(c).items()}
            solve_fn(columns[:, lo:hi], views)
        pending += gather_stripe(buffers, c, gather, group, async_op=True)
    for w in pending:
        w.wait()
    return buffers.result()
