"""Batch-axis sharding over the GPUs of one node: one process per GPU (torch.distributed, backend "nccl" = RCCL
over xGMI on ROCm; "gloo" for the CPU tests), contiguous slices of the leading batch dimension per rank, NO
collective inside the solve, one all_gather of the results at the end.

The reference has no multi-GPU code; batch members are independent in every operation of the path, so this is
the whole distributed design.  The reference's stopping rules are global over the batch (mean residual,
linear_cg.py:304; shared pivoted-Cholesky rank, _pivoted_cholesky.py:57).  By default each shard applies them to its
own members ("option B": identical whenever CG ends at the iteration floors -- all BASELINE configs except the long
Kronecker run); inside `global_stopping_rule(group)` three doubles are all-reduced per stopping-rule evaluation and
the sharded run executes exactly the iterations of the unsharded one ("option A").

Operators that do not fit one GPU (BASELINE cfg5: 256 x 16384^2 fp32 = 256 GiB) are never materialised whole:
`build_local_shard` / `sharded_solve_from_factory` call a user factory with this rank's [lo, hi) member range and
build only that slice, on this rank's device.
"""
from __future__ import annotations

import contextlib
import threading
from typing import Callable, Optional

import torch
import torch.distributed as dist

_local = threading.local()


class StopReduce:
    """All-reduce (SUM) of the stopping-rule statistic {sum of residual norms, number of columns, abort request} over
    the ranks of `group`: three doubles per evaluation, the only traffic the batch-global rule adds (SURVEY 8(e) option A)."""

    def __init__(self, group=None, device: Optional[torch.device] = None):
        self.group = group
        backend = dist.get_backend(group)
        self.device = device if device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))
        self.calls = 0

    def __call__(self, vals):
        t = torch.tensor(list(vals), dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        self.calls += 1
        return t.tolist()


def active_stop_reduce():
    return getattr(_local, "stop_reduce", None)


@contextlib.contextmanager
def global_stopping_rule(group=None, reducer: Optional[Callable] = None):
    """Inside this context every `linear_cg` call evaluates the reference's stopping rule on the statistic of ALL ranks
    of `group` (the mean residual over the whole sharded batch, linear_cg.py:302-308) and the pivoted-Cholesky
    preconditioner takes the rank every shard would take together (_pivoted_cholesky.py:57): sharded runs then
    execute exactly the iterations the unsharded run executes.  Outside it ("option B") each shard applies the rule to
    its own members -- identical whenever CG ends at its iteration floors.  `reducer`: custom callable(list[3]) ->
    list[3] (tests)."""
    prev, prev_group = getattr(_local, "stop_reduce", None), getattr(_local, "group", None)
    _local.stop_reduce = reducer if reducer is not None else StopReduce(group)
    _local.group = group
    try:
        yield _local.stop_reduce
    finally:
        _local.stop_reduce, _local.group = prev, prev_group


def global_max_int(value: int) -> int:
    """MAX over the ranks of the active global-rule context (identity outside one)."""
    red = active_stop_reduce()
    if red is None:
        return int(value)
    if isinstance(red, StopReduce):
        t = torch.tensor([int(value)], dtype=torch.int64, device=red.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=red.group)
        return int(t.item())
    return int(getattr(red, "max_int", lambda v: v)(int(value)))


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


# ---- operators that do not fit one GPU: construct only this rank's slice ------------------------------------------
def build_local_shard(factory: Callable, total: int, group=None):
    """factory(lo, hi) -> (operator, rhs) for the members [lo, hi) of a batch of `total`, built directly on this rank's
    device.  Returns (operator_shard, rhs_shard, (lo, hi)).  Nothing outside the slice is ever allocated."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(total, rank, world)
    op_s, rhs_s = factory(lo, hi)
    if len(op_s.batch_shape) == 0 or op_s.batch_shape[0] != hi - lo or rhs_s.shape[0] != hi - lo:
        raise ValueError(f"factory({lo}, {hi}) must return {hi - lo} members, got operator batch {tuple(op_s.batch_shape)}")
    return op_s, rhs_s, (lo, hi)


def sharded_solve_from_factory(factory: Callable, total: int, group=None, solve_fn: Optional[Callable] = None,
                               gather: bool = True, global_rule: bool = False):
    """A^{-1} rhs with the batch split across ranks, every rank constructing only its own members (see
    build_local_shard).  gather=False returns the local solutions (the caller keeps them sharded).
    global_rule=True evaluates the stopping rules over the whole batch (global_stopping_rule)."""
    op_s, rhs_s, _ = build_local_shard(factory, total, group)
    ctx = global_stopping_rule(group) if global_rule else contextlib.nullcontext()
    with ctx:
        x_s = solve_fn(op_s, rhs_s) if solve_fn is not None else op_s.solve(rhs_s)
    return all_gather_batch(x_s, total, group) if gather else x_s
