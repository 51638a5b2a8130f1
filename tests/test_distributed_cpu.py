"""world_size-2 `gloo` tests of the batch-sharding layer (linear_operator_amd/distributed.py) on CPU.
The product solve needs the HIP extension, so the per-shard compute is injected here: the ORACLE plays the
solver (test infrastructure as the checker); what is under test is slicing the operator tree, uneven shards,
and the single all_gather at the end -- the result must equal the unsharded oracle solve bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(fn, B, out_dir):
    """Two gloo ranks on a free local port; ONE retry on a fresh port when the rendezvous itself fails (the port found free
    can be taken between its probe and the store's bind -- seen once in a few dozen runs of this suite)."""
    for attempt in (0, 1):
        try:
            mp.spawn(fn, args=(2, _free_port(), B, out_dir), nprocs=2, join=True)
            return
        except Exception as e:  # noqa: BLE001
            msg = str(e)
            if attempt or not any(k in msg for k in ("address already in use", "Address already in use", "EADDRINUSE",
                                                     "Connection refused", "connect() timed out", "store")):
                raise


def _worker(rank, world, port, B, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cases
    from linear_operator_amd import distributed as D
    from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
    from oracle import lo_oracle as orc

    C, d, rhs = cases.lowrank_diag(777, B, 96, 4, 2)
    A = AddedDiagLinearOperator(LowRankRootLinearOperator(torch.from_numpy(C)), DiagLinearOperator(torch.from_numpy(d)))

    def oracle_solve(op_s, rhs_s):
        Cs, ds = (t.numpy() for t in op_s.representation())
        x, _, _ = orc.linear_cg(lambda v: orc.matvec_lowrank_diag(Cs, ds, v), rhs_s.numpy(), tolerance=1e-4)
        return torch.from_numpy(x)

    def oracle_iql(op_s, rhs_s):
        Cs, ds = (t.numpy() for t in op_s.representation())
        x, _, _ = orc.linear_cg(lambda v: orc.matvec_lowrank_diag(Cs, ds, v), rhs_s.numpy(), tolerance=1e-4)
        dense = Cs @ np.swapaxes(Cs, -1, -2) + np.stack([np.diag(z) for z in ds])
        return torch.from_numpy((x * rhs_s.numpy()).sum(-2).sum(-1)), torch.from_numpy(
            np.linalg.slogdet(dense)[1].astype(np.float32))

    lo_, hi_ = D.shard_bounds(B, rank, world)
    shard = D.shard_operator(A, rank, world)
    assert shard.shape == (hi_ - lo_, 96, 96) and type(shard) is AddedDiagLinearOperator
    x = D.sharded_solve(A, torch.from_numpy(rhs), solve_fn=oracle_solve)
    iq, ld = D.sharded_inv_quad_logdet(A, torch.from_numpy(rhs), fn=oracle_iql)
    if rank == 0:
        np.savez(os.path.join(out_dir, f"res_{B}.npz"), x=x.numpy(), iq=iq.numpy(), ld=ld.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [6, 5])  # even and uneven shards
def test_sharded_solve_gloo_world2(tmp_path, B):
    import cases
    from oracle import lo_oracle as orc

    _spawn(_worker, B, str(tmp_path))
    got = np.load(tmp_path / f"res_{B}.npz")
    C, d, rhs = cases.lowrank_diag(777, B, 96, 4, 2)
    # per-shard stopping rule == global rule here (ends at the 11-iteration floor), so results are identical
    parts = []
    for lo_, hi_ in [(0, (B + 1) // 2), ((B + 1) // 2, B)]:
        x, _, _ = orc.linear_cg(lambda v: orc.matvec_lowrank_diag(C[lo_:hi_], d[lo_:hi_], v), rhs[lo_:hi_],
                                tolerance=1e-4)
        parts.append(x)
    assert np.array_equal(got["x"], np.concatenate(parts, 0))
    full, _, _ = orc.linear_cg(lambda v: orc.matvec_lowrank_diag(C, d, v), rhs, tolerance=1e-4)
    assert np.allclose(got["x"], full, rtol=1e-5, atol=1e-6)
    assert got["iq"].shape == (B,) and got["ld"].shape == (B,)


def test_shard_bounds_cover_batch():
    from linear_operator_amd.distributed import shard_bounds

    for B in (1, 7, 8, 512, 1023):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _worker_factory(rank, world, port, B, out_dir):
    """Factory sharding (only this rank's members are ever built) and the plumbing of the batch-global stopping rule:
    the oracle plays the per-shard solver and calls the active reducer the way the device engine does."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cases
    from linear_operator_amd import distributed as D
    from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
    from oracle import lo_oracle as orc

    C, d, rhs = cases.lowrank_diag(778, B, 96, 4, 2)
    built = []

    def factory(lo_, hi_):
        built.append((lo_, hi_))
        op = AddedDiagLinearOperator(LowRankRootLinearOperator(torch.from_numpy(C[lo_:hi_])),
                                     DiagLinearOperator(torch.from_numpy(d[lo_:hi_])))
        return op, torch.from_numpy(rhs[lo_:hi_])

    seen = {}

    def oracle_solve(op_s, rhs_s):
        Cs, ds = (t.numpy() for t in op_s.representation())
        x, _, info = orc.linear_cg(lambda v: orc.matvec_lowrank_diag(Cs, ds, v), rhs_s.numpy(), tolerance=1e-4)
        red = D.active_stop_reduce()
        seen["reducer"] = red
        if red is not None:  # the statistic the engine hands over: local residual sum, local column count, abort flag
            n = float(rhs_s.shape[0] * rhs_s.shape[-1])
            seen["reduced"] = red([info.mean_residual * n, n, 0.0])
            seen["m"] = D.global_max_int(10 + rank)
        return torch.from_numpy(x)

    x = D.sharded_solve_from_factory(factory, B, solve_fn=oracle_solve, global_rule=True)
    lo_, hi_ = D.shard_bounds(B, rank, world)
    assert built == [(lo_, hi_)], "only this rank's slice may be built"
    assert isinstance(seen["reducer"], D.StopReduce) and D.active_stop_reduce() is None
    assert abs(seen["reduced"][1] - B * 2) < 1e-9 and seen["reduced"][2] == 0.0  # counts of both ranks added up
    assert seen["m"] == 10 + world - 1
    x_local = D.sharded_solve_from_factory(factory, B, solve_fn=oracle_solve, gather=False)
    assert x_local.shape[0] == hi_ - lo_ and seen["reducer"] is None
    if rank == 0:
        np.savez(os.path.join(out_dir, f"fac_{B}.npz"), x=x.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [6, 5])
def test_factory_sharding_and_global_rule_plumbing_gloo_world2(tmp_path, B):
    import cases
    from oracle import lo_oracle as orc

    _spawn(_worker_factory, B, str(tmp_path))
    got = np.load(tmp_path / f"fac_{B}.npz")
    C, d, rhs = cases.lowrank_diag(778, B, 96, 4, 2)
    full, _, _ = orc.linear_cg(lambda v: orc.matvec_lowrank_diag(C, d, v), rhs, tolerance=1e-4)
    assert got["x"].shape == full.shape and np.allclose(got["x"], full, rtol=1e-5, atol=1e-6)


def _run_bench_stub(extra_env, argv):
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(LO_BENCH_STUB="1", **extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout  # exactly ONE JSON line on stdout (the driver's contract)
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must start its own 2-rank job (the round-2
    script exited instead): stubbed solver on gloo, so the launcher, the rank plumbing, the barrier + MAX-over-ranks
    timing and the JSON contract run here without a GPU."""
    out = _run_bench_stub({}, ["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert len(out["per_rank_ms"]) == 2 and out["gather_ok"]
    assert out["ms_per_step"] == max(out["per_rank_ms"])  # the slowest rank is the job's time
    assert out["per_rank_ms"][1] > out["per_rank_ms"][0] * 0.9  # (rank 1 sleeps twice as long per step)
    assert out["data"] == "stub" and out["metric"] == "stub"  # a stub line can never pass for a measurement


def test_bench_single_rank_contract_and_world_size_mismatch():
    out = _run_bench_stub({}, ["--steps", "2", "--warmup", "1"])
    assert out["n_gpus"] == 1 and len(out["per_rank_ms"]) == 1
    import subprocess

    env = dict(os.environ, LO_BENCH_STUB="1", WORLD_SIZE="1", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr  # a launcher that started the wrong number of ranks
