"""Batch-axis sharding over the GPUs of one node: one process per GPU (torch.distributed, backend "nccl" = RCCL
over xGMI on ROCm; "gloo" for the CPU tests), contiguous slices of the leading batch dimension per rank, NO
collective inside the solve, one all_gather of the results at the end.

The reference has no multi-GPU code; batch members are independent in every operation of the path, so this is
the whole distributed design.  Caveat documented in DESIGN.md: the reference's stopping rules are global over
the batch (mean residual, linear_cg.py:304); each shard applies the rule to its own members ("option B"), which
is identical whenever CG ends at the iteration floors (all BASELINE configs except the long Kron run).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist


def shard_bounds(batch: int, rank: int, world: int):
    """Contiguous, balanced [lo, hi) slice of `batch` members for `rank` (first batch % world ranks get +1)."""
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_tensor(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_bounds(t.shape[0], rank, world)
    return t[lo:hi]


def shard_operator(op, rank: int, world: int):
    """Slice every leaf tensor of the operator tree along its leading (batch) dimension."""
    leaves = op.representation()
    B = op.batch_shape[0] if len(op.batch_shape) else None
    if B is None:
        raise ValueError("shard_operator needs a batched operator (leading batch dimension)")
    for t in leaves:
        if t.dim() == 0 or t.shape[0] != B:
            raise ValueError(f"leaf tensor of shape {tuple(t.shape)} does not carry the batch dimension {B} first")
    return op.representation_tree()(*[shard_tensor(t, rank, world) for t in leaves])


def all_gather_batch(x: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """Concatenate per-rank shards (possibly of unequal size) along dim 0 -> [total, ...] on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_bounds(total, r, world) for r in range(world)]
    counts = [hi - lo for lo, hi in sizes]
    assert x.shape[0] == counts[rank], (x.shape, counts, rank)
    if len(set(counts)) == 1:
        out = torch.empty(total, *x.shape[1:], dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
        return out
    m = max(counts)
    padded = torch.zeros(m, *x.shape[1:], dtype=x.dtype, device=x.device)
    padded[: x.shape[0]] = x
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def sharded_solve(op, rhs: torch.Tensor, group=None, solve_fn: Optional[Callable] = None) -> torch.Tensor:
    """A^{-1} rhs for a batched operator, batch members split across the ranks of `group`.
    Every rank passes the FULL operator / rhs (or at least its own slice would do); returns the full result."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    total = rhs.shape[0]
    op_s = shard_operator(op, rank, world)
    rhs_s = shard_tensor(rhs, rank, world)
    x_s = solve_fn(op_s, rhs_s) if solve_fn is not None else op_s.solve(rhs_s)
    return all_gather_batch(x_s, total, group)


def sharded_inv_quad_logdet(op, inv_quad_rhs: torch.Tensor, group=None, fn: Optional[Callable] = None):
    """(inv_quad [B], logdet [B]) with the batch split across ranks; only the scalars are gathered."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    total = inv_quad_rhs.shape[0]
    op_s = shard_operator(op, rank, world)
    rhs_s = shard_tensor(inv_quad_rhs, rank, world)
    iq, ld = fn(op_s, rhs_s) if fn is not None else op_s.inv_quad_logdet(rhs_s, logdet=True)
    return all_gather_batch(iq.reshape(-1), total, group), all_gather_batch(ld.reshape(-1), total, group)
