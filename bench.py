#!/usr/bin/env python3
"""bench.py -- throughput of the batched preconditioned-CG hot path on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver
launches it under torch.distributed.run, one rank per GPU.  Prints ONE JSON line on rank 0.

Workload (config.workload): BASELINE.json north_star headline -- a batch of 512 operators
A_b = C_b C_b^T + diag(d_b), C [512, 8192, 32], d [512, 8192] (AddedDiag(LowRankRoot, Diag)), one right-hand
side column, solved by `linear_cg` with the reference's rank-15 pivoted-Cholesky preconditioner at
cg_tolerance 1e-4 (stops at the 11-iteration floor).  A "step" is one full linear_cg call
over the whole batch (init, 11 iterations, un-normalise; preconditioner already built, as the reference's
`_solve` receives it).  Inputs are resident in HBM before the timed region.
  metric  = member-matvecs per second inside CG = batch members x the operator applications linear_cg performs (11) / CG
            wall time -- CG-ITERATION EQUIVALENTS: the timed engine (k_cg_rspace3) reads the operator once per solve and
            runs the iterations on R + 1 Krylov coordinates (`value_kind` says so in the line).  Beside it, first level:
            `operator_applications_per_sec` (the engine that performs the 11 products on the resident rows) and
            `hbm_matvec` (ONE application of the operator as an HBM pass, lo_matvec_f32 -> k_lr_mv: north_star's "batched
            CG matvec" with its roofline fraction by SURVEY 8(d)'s bytes)
  value   = whole-job aggregate over all ranks (weak scaling: every rank owns 512 members; the solutions are
            all-gathered over RCCL when N > 1, as north_star prescribes: --gather end | step)
Also reported: end-to-end solves/s (preconditioner build included), what a solve costs when its preconditioner cache
serves 1 / 25 / 100 solves, the roofline line of the dominant kernel (live HIP-event timing on the launch stream), and
the CPU oracle timed on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from linear_operator_amd import _hip  # noqa: E402
from linear_operator_amd import kernels as K  # noqa: E402

B_PER_GPU, N, R, C_COLS, RANK_K = 512, 8192, 32, 1, 15
TOL = 1e-4
ITERS_FLOOR = 11  # linear_cg.py:303 -- the iterations the operator-resident kernel runs in one launch
PROFILE_DIR = "r06"  # profiles/<dir>/ holds the rocprofv3 summaries of this command (tools/profile_round.sh)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6300.0  # the guide's measured float4-copy rate: the ceiling a streaming kernel can reach
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak (v_mfma_f32_16x16x4_f32 / 32x32x2, MI355X_MICROARCH.md)


def make_problem(device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    Cm = torch.randn(B_PER_GPU, N, R, generator=g, device=device, dtype=torch.float32) / (R ** 0.5)
    d = torch.rand(B_PER_GPU, N, generator=g, device=device, dtype=torch.float32) + 0.5
    rhs = torch.randn(B_PER_GPU, N, C_COLS, generator=g, device=device, dtype=torch.float32)
    return Cm, d, rhs


def build_precond(desc, d, need_q=True):
    """Rank-15 pivoted-Cholesky preconditioner of the low-rank operator, in the generic (Q, 1/d) form of the reference's
    cache and in root form (F = M (I + M^T E M)^-1 M^T: what the operator-resident CG kernels use)."""
    L, perm = K.pivoted_cholesky(desc, RANK_K, contiguous=False)  # [B, m, N] rows read in place by the build
    return K.precond_build(L, d, constant_diag=False, root=desc.A0, perm=perm, need_q=need_q)


def algorithmic_bytes(name, k_eff):
    """Compulsory HBM bytes of one launch of a kernel class (DESIGN.md section 'kernels'), fp32."""
    B, c = B_PER_GPU, C_COLS
    if name.startswith("skinny_tn_R"):
        r = R if name.endswith("R32") else k_eff
        return 4 * B * (N * r + N * c)  # stream A once + the vector; partial outputs are KB
    if name.startswith("skinny_nn_R"):
        r = R if name.endswith("R32") else k_eff
        return 4 * B * (N * r + N + 2 * N * c)  # stream A once + diagonal + vector in + vector out
    if name == "cg_onchip":
        # operator-resident CG: ONE launch runs all guaranteed iterations and reads the operator ONCE.  Compulsory
        # bytes per launch with the root-form preconditioner (no second tall matrix): C, d, 1/d, the right-hand side in
        # and the solution out, plus the two R x R matrices per member
        return 4 * B * (N * (R + 2 + 2 * c) + 2 * R * R)
    if name == "cg_update_xr":
        return 4 * B * N * c * 6  # r, Ap, x, p read; r, x written
    if name == "cg_update_p":
        return 4 * B * N * c * 3
    if name == "vec_dot_part":
        return 4 * B * N * c * 2
    return 0


def woodbury_fp64_rel_err(Cm, d, rhs, x, chunk=64):
    """max_b ||x_b - x*_b|| / ||x*_b|| with x* = (D + C C^T)^-1 rhs by the Woodbury identity in fp64 (checker only)."""
    worst = 0.0
    for b0 in range(0, Cm.shape[0], chunk):
        C64, d64 = Cm[b0:b0 + chunk].double(), d[b0:b0 + chunk].double().unsqueeze(-1)
        r64 = rhs[b0:b0 + chunk].double()
        Cd = C64 / d64
        cap = torch.eye(C64.shape[-1], device=Cm.device, dtype=torch.float64) + C64.mT @ Cd
        xs = r64 / d64 - Cd @ torch.linalg.solve(cap, C64.mT @ (r64 / d64))
        err = (x[b0:b0 + chunk].double() - xs).flatten(1).norm(dim=1) / xs.flatten(1).norm(dim=1)
        e = float(err.max().item())
        if e != e:  # NaN must not hide behind max()
            return float("nan")
        worst = max(worst, e)
    return worst


def cpu_baseline(seconds_budget=10.0, device=None):
    """The CPU restatement of the path (kind 'port') timed on the host cores on a bounded sample of the same workload:
    the C oracle (oracle/lo_oracle_c.c: linear_cg with the rank-15 pivoted-Cholesky preconditioner, OpenMP over the
    members, every core) on 128 of the 512 members; the numpy oracle's rate on 8 members beside it.  The same leg also
    holds the parity sample of the line (`parity_sample`): the HIP path against the numpy oracle on IDENTICAL inputs --
    solve rel-err, logdet rel-err with identical probes and preconditioner (BASELINE.json's metric names it), pivots."""
    import cases
    import numpy as np
    from oracle import lo_oracle as orc
    from oracle import lo_oracle_c as occ

    Bc = 128
    Cc, dc, rc_ = cases.lowrank_diag(4242, Bc, N, R, C_COLS)
    Bs = 8
    Cs, ds, rs = Cc[:Bs], dc[:Bs], rc_[:Bs]
    # ---- C oracle, all cores ----
    op_c = occ.lowrank_diag(Cc, dc)
    Lc, piv_c = occ.pivoted_cholesky(occ.lowrank_diag(Cc), RANK_K)
    pre_c = occ.Preconditioner(Lc, dc)
    x_c, _, info_c = occ.linear_cg(op_c, rc_, pre=pre_c, tolerance=TOL)  # warm up
    t0 = time.perf_counter()
    reps_c, mv_c = 0, 0
    while time.perf_counter() - t0 < seconds_budget:
        _, _, info_c = occ.linear_cg(op_c, rc_, pre=pre_c, tolerance=TOL)
        mv_c += Bc * info_c.matvecs
        reps_c += 1
    dt_c = time.perf_counter() - t0
    # ---- numpy oracle (the restatement the parity tests use), a few seconds ----
    L, piv = orc.pivoted_cholesky(orc.LowRankRowSource(Cs), RANK_K)
    pre = orc.Preconditioner(L, ds)
    mm = lambda v: orc.matvec_lowrank_diag(Cs, ds, v)  # noqa: E731
    x_ref, _, _ = orc.linear_cg(mm, rs, tolerance=TOL, preconditioner=pre.apply)  # warm up
    t0 = time.perf_counter()
    reps, mv = 0, 0
    while time.perf_counter() - t0 < 3.0:
        _, _, info = orc.linear_cg(mm, rs, tolerance=TOL, preconditioner=pre.apply)
        mv += Bs * info.matvecs
        reps += 1
    dt = time.perf_counter() - t0
    c_vs_numpy = float(np.max(np.linalg.norm((x_c[:Bs] - x_ref).reshape(Bs, -1), axis=1) /
                              np.linalg.norm(x_ref.reshape(Bs, -1), axis=1)))
    out = {"value": mv_c / dt_c, "unit": "member-matvecs/s", "cores": occ.num_threads(), "kind": "port",
           "sample": f"{reps_c} x C oracle (oracle/lo_oracle_c.c, OpenMP over members) linear_cg on {Bc} of the "
                     f"{B_PER_GPU} members (same N={N}, R={R}, c={C_COLS}, rank-{RANK_K} pivoted-Cholesky preconditioner, "
                     f"tol {TOL}) in {dt_c:.1f} s; the reference counts {11 + 1} products per solve (it also spends one "
                     "on A x0)",
           "numpy_oracle_value": mv / dt,
           "numpy_oracle_sample": f"{reps} x numpy oracle linear_cg on {Bs} members in {dt:.1f} s",
           "c_vs_numpy_oracle_solve_rel_err": c_vs_numpy,
           "c_vs_numpy_oracle_pivots_equal": bool(np.array_equal(piv_c[:Bs], np.asarray(piv)))}
    if device is not None:  # ---- parity sample: the HIP path on the oracle's very inputs ----
        Pn = 16
        Ct, dt_, rt = (torch.from_numpy(a).to(device) for a in (Cs, ds, rs))
        desc = K.lowrank_diag_descriptor(Ct, dt_)
        Lh, ph = K.pivoted_cholesky(desc, RANK_K, contiguous=False)
        pivots_equal = bool(np.array_equal(ph.cpu().numpy()[:, :RANK_K], np.asarray(piv)[:, :RANK_K]))
        pre_h = K.precond_build(Lh, dt_, constant_diag=False, root=desc.A0, perm=ph)
        xh = K.cg_solve(desc, rt, precond=pre_h, tolerance=TOL).x.cpu().numpy()
        num = np.linalg.norm((xh - x_ref).reshape(Bs, -1), axis=1)
        solve_err = float((num / np.linalg.norm(x_ref.reshape(Bs, -1), axis=1)).max())
        rng = np.random.Generator(np.random.PCG64(99))
        Z = rng.standard_normal((Bs, N, Pn)).astype(np.float32)
        Z /= np.linalg.norm(Z, axis=-2, keepdims=True)
        full = np.concatenate([Z, rs], axis=-1)
        xo, to, _ = orc.linear_cg(mm, full, n_tridiag=Pn, tolerance=TOL, preconditioner=pre.apply)
        ev, evec = orc.lanczos_tridiag_to_diag(to)
        ld_ref = orc.slq_logdet(N, ev, evec) + pre.logdet
        rh = K.cg_solve(desc, torch.from_numpy(full).to(device), precond=pre_h, n_tridiag=Pn, tolerance=TOL)
        _, _, ldh = K.tridiag_eigh_slq(rh.t_mat, N)
        ld_h = (ldh + pre_h.logdet.reshape(-1)).cpu().numpy()
        ld_exact = orc.woodbury_logdet(Cs.astype(np.float64), ds.astype(np.float64))
        out["parity_sample"] = {
            "members": Bs, "pivots_equal": pivots_equal, "solve_rel_err_vs_oracle": solve_err,
            "logdet_rel_err_vs_oracle": float(np.max(np.abs(ld_h - ld_ref) / np.abs(ld_ref))),
            "logdet_rel_err_slq_vs_exact_fp64": float(np.max(np.abs(ld_h - ld_exact) / np.abs(ld_exact))),
            "note": "HIP path vs the numpy oracle on identical inputs: identical right-hand side, identical 16 probes, "
                    "identical rank-15 preconditioner (pivots compared exactly); logdet = SLQ on the CG tridiagonals "
                    "+ logdet P.  The last figure is the estimator's own error against the fp64 closed form "
                    "(16 probes), not a parity figure"}
    return out


def config_baselines(device):
    """SURVEY 8(d) for the other BASELINE configs: for cfg3, the cfg4 shard and the cfg5 shard a `cpu_baseline` (the C
    oracle, oracle/lo_oracle_c.c, OpenMP over the members, timed on this box's host cores at the SAME per-member shape on
    a host-sized batch) and a `parity` block (the HIP path on the oracle's very inputs: pivots compared exactly, solve /
    inv_quad / logdet relative errors).  The oracle is the checker and the baseline here, never the thing measured."""
    import cases
    import numpy as np
    from oracle import lo_oracle_c as occ

    cores = occ.num_threads()
    out = {}

    def rel_cols(x, ref):
        num = np.sqrt(((x.astype(np.float64) - ref) ** 2).sum(-2))
        return float((num / np.maximum(np.sqrt((ref.astype(np.float64) ** 2).sum(-2)), 1e-300)).max())

    def timed(fn, budget):
        fn()  # warm (page faults, OpenMP team)
        t0, reps = time.perf_counter(), 0
        while True:
            r = fn()
            reps += 1
            if time.perf_counter() - t0 >= budget or reps >= 50:
                return (time.perf_counter() - t0) / reps, reps, r

    # ---- cfg3: 16 probes + 1 right-hand side, rank-15 preconditioner, tridiagonals + SLQ (inv_quad_logdet) ----
    Bc = max(8, min(32, cores))
    C3, d3, r3 = cases.lowrank_diag(4301, Bc, N, R, 1)
    Z3, _ = cases.probes(4302, Bc, N, 16)
    t_c, reps, (iq_o, ld_o, x_o, t_o, info_o, piv_o) = timed(
        lambda: occ.inv_quad_logdet(occ.lowrank_diag(C3, d3), occ.lowrank_diag(C3), d3, r3, Z3, tolerance=TOL), 3.0)
    Ct, dt_ = torch.from_numpy(C3).to(device), torch.from_numpy(d3).to(device)
    desc = K.lowrank_diag_descriptor(Ct, dt_)
    pre = build_precond(desc, dt_)
    full = torch.from_numpy(np.concatenate([Z3, r3], -1)).to(device)
    rh = K.cg_solve(desc, full, precond=pre, n_tridiag=16, tolerance=TOL)
    _, _, slq = K.tridiag_eigh_slq(rh.t_mat, N)
    ld_h = (slq + pre.logdet.reshape(-1)).cpu().numpy()
    xh = rh.x.cpu().numpy()
    iq_h = (xh[..., 16:] * r3).sum(-2)
    L3, p3 = K.pivoted_cholesky(desc, RANK_K)
    out["cfg3"] = {
        "cpu_baseline": {"value": Bc / t_c, "unit": "member-solve-logdets/s", "cores": cores, "kind": "port",
                         "sample": f"{reps} x C oracle inv_quad_logdet (pivoted Cholesky + preconditioner + linear_cg with "
                                   f"16 tridiagonals + eigh / SLQ) on {Bc} members of the cfg3 shape in {t_c * reps:.1f} s"},
        "parity": {"members": Bc, "pivots_equal": bool(np.array_equal(p3.cpu().numpy()[:, :RANK_K], piv_o[:, :RANK_K])),
                   "iterations": [int(rh.iterations), int(info_o.iterations)],
                   "solve_rel_err_vs_oracle": rel_cols(xh, x_o),
                   "inv_quad_rel_err_vs_oracle": float(np.max(np.abs(iq_h - iq_o) / np.abs(iq_o))),
                   "logdet_rel_err_vs_oracle": float(np.max(np.abs(ld_h - ld_o) / np.abs(ld_o))),
                   "note": "identical right-hand side, identical 16 probes; HIP path = rank-15 pivoted Cholesky, root-form "
                           "preconditioner, CG with tridiagonals (R-space columns), fp64 QL + SLQ on the device"}}
    del Ct, dt_, desc, pre, full, rh
    # ---- cfg4 shard: Kronecker 256 (x) 256 + 1e-2 I, one column, tolerance 1e-3 ----
    B4 = max(4, min(16, cores))
    K1, K2, sig, r4 = cases.kron_factors(4401, B4, 256, 256, 1)

    def cfg4_cpu():
        L, piv = occ.pivoted_cholesky(occ.kron_diag(K1, K2), RANK_K)
        pre_c = occ.Preconditioner(L, sig, const_diag=True)
        x, _, info = occ.linear_cg(occ.kron_diag(K1, K2, sig, const_diag=True), r4, pre=pre_c, tolerance=1e-3)
        return x, info, piv

    t0 = time.perf_counter()
    x4o, info4, piv4 = cfg4_cpu()
    t_c4 = time.perf_counter() - t0
    K1t, K2t, sigt = (torch.from_numpy(a).to(device) for a in (K1, K2, sig.reshape(-1)))
    desc4 = K.kron_diag_descriptor(K1t, K2t, sigt, const_diag=True)
    L4, p4 = K.pivoted_cholesky(desc4.without_diag(), RANK_K, contiguous=False)
    r4h = K.cg_solve(desc4, torch.from_numpy(r4).to(device), precond=K.precond_build(L4, sigt, True, perm=p4, kron=desc4),
                     tolerance=1e-3)
    out["cfg4_shard"] = {
        "cpu_baseline": {"value": B4 / t_c4, "unit": "member-solves/s", "cores": cores, "kind": "port",
                         "sample": f"1 x C oracle (pivoted Cholesky of the Kronecker part + preconditioner + linear_cg to "
                                   f"tolerance 1e-3, {info4.iterations} iterations) on {B4} members of the cfg4 shape "
                                   f"(256 (x) 256 + 1e-2 I) in {t_c4:.1f} s"},
        "parity": {"members": B4, "pivots_equal": bool(np.array_equal(p4.cpu().numpy()[:, :RANK_K], piv4[:, :RANK_K])),
                   "iterations": [int(r4h.iterations), int(info4.iterations)],
                   "solve_rel_err_vs_oracle": rel_cols(r4h.x.cpu().numpy(), x4o),
                   "note": "both stop on the batch-global rule mean ||r|| < 1e-3 (linear_cg.py:302-308); with different "
                           "summation orders the crossing can fall one iteration apart, the solutions then differ at the "
                           "level of the tolerance -- the iteration-pinned golden g24 carries the 1e-4 bar"}}
    del K1t, K2t, desc4, L4, r4h
    # ---- cfg5 shard: dense 16384^2 + diag, 16 probes + 1 right-hand side ----
    B5, N5 = 2, 16384
    rng = np.random.Generator(np.random.PCG64(4501))
    Y = (rng.standard_normal((B5, N5, 256)) / 16).astype(np.float32)
    K5 = Y @ np.swapaxes(Y, -1, -2)
    del Y
    d5 = (rng.random((B5, N5)) + 0.5).astype(np.float32)
    r5 = rng.standard_normal((B5, N5, 1)).astype(np.float32)
    Z5, _ = cases.probes(4502, B5, N5, 16)
    t0 = time.perf_counter()
    iq5, ld5, x5, _, info5, piv5 = occ.inv_quad_logdet(occ.dense_diag(K5, d5), occ.dense_diag(K5), d5, r5, Z5, tolerance=TOL)
    t_c5 = time.perf_counter() - t0
    K5t, d5t = torch.from_numpy(K5).to(device), torch.from_numpy(d5).to(device)
    desc5 = K.dense_diag_descriptor(K5t, d5t)
    L5, p5 = K.pivoted_cholesky(desc5, RANK_K, contiguous=False)
    pre5 = K.precond_build(L5, d5t, False)
    r5h = K.cg_solve(desc5, torch.from_numpy(np.concatenate([Z5, r5], -1)).to(device), precond=pre5, n_tridiag=16,
                     tolerance=TOL)
    _, _, slq5 = K.tridiag_eigh_slq(r5h.t_mat, N5)
    ld5h = (slq5 + pre5.logdet.reshape(-1)).cpu().numpy()
    x5h = r5h.x.cpu().numpy()
    out["cfg5_shard"] = {
        "cpu_baseline": {"value": B5 / t_c5, "unit": "member-solve-logdets/s", "cores": min(cores, B5), "kind": "port",
                         "sample": f"1 x C oracle inv_quad_logdet on {B5} members of 16384^2 (K = Y Y^T, Y [16384, 256], + "
                                   f"diag; the oracle parallelises over members: {min(cores, B5)} cores busy) in "
                                   f"{t_c5:.1f} s, {info5.iterations} iterations; per member, not scaled"},
        "parity": {"members": B5, "pivots_equal": bool(np.array_equal(p5.cpu().numpy()[:, :RANK_K], piv5[:, :RANK_K])),
                   "iterations": [int(r5h.iterations), int(info5.iterations)],
                   "solve_rel_err_vs_oracle": rel_cols(x5h, x5),
                   "inv_quad_rel_err_vs_oracle": float(np.max(np.abs((x5h[..., 16:] * r5).sum(-2) - iq5) / np.abs(iq5))),
                   "logdet_rel_err_vs_oracle": float(np.max(np.abs(ld5h - ld5) / np.abs(ld5)))}}
    return out


def _gc_off():
    """Every timed region of this process runs with the cyclic collector off, switched off ONCE before the first
    warm-up: a generation-2 pass walks ~1e6 objects for 35 ms -- 80 headline steps -- and where it falls depends on the
    allocation count up to that point (seen: the first timed repetition of the 64-member solve, 36.9 ms instead of
    0.29); collecting right in front of a timed region is no alternative, the GPU's clocks drop during the pause and
    the following repetitions run 3 - 5 % slow."""
    import gc

    gc.collect()
    gc.disable()


def _time(fn, reps, warm_seconds=0.02, min_seconds=0.05, warm_calls=1):
    """Seconds per call of fn.  At least `warm_calls` untimed calls and >= 20 ms of them (the first calls after a pause of
    the device run 3 - 7 % slow, see the soak of the headline), then `reps` calls -- more if those take less than 50 ms.
    (warm_calls > 1: paths whose FIRST call takes a second -- autograd graphs, allocator growth -- and whose second still
    allocates: one warm call measured 22 ms per step where the steady state is 7.5.)"""
    t_end = time.perf_counter() + warm_seconds
    done = 0
    while True:
        fn()
        torch.cuda.synchronize()
        done += 1
        if time.perf_counter() >= t_end and done >= warm_calls:
            break
    while True:
        out = None
        t0 = time.perf_counter()
        for _ in range(reps):
            out = None  # (a result of several GB must not be alive while the next call allocates its own)
            out = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dt >= min_seconds or reps >= 4096:
            return dt / reps, out
        reps = min(4096, max(reps + 1, int(reps * min_seconds / max(dt, 1e-6)) + 1))


def _profiled(fn):
    """Run fn once more with the library's HIP-event timers on (events on the launch stream): {name: (launches, ms)}."""
    torch.cuda.synchronize()
    _hip.prof_enable(True)
    try:
        fn()
        torch.cuda.synchronize()
        return _hip.prof_report()
    finally:
        _hip.prof_enable(False)


def _committed_traffic(fname, key=None):
    """(bytes per launch, source string) of a kernel from the committed rocprofv3 counter summary, or (None, None)."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", PROFILE_DIR, fname)))
        ent = tj["kernels"][key] if key else tj
        return float(ent["traffic_bytes_per_launch"]), (
            f"profiles/{PROFILE_DIR}/{fname}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, kernel {ent['kernel']}, "
            "FETCH x2 (gfx950 correction) + WRITE")
    except Exception:  # noqa: BLE001 -- no committed profile for this kernel: no traffic figure
        return None, None


def _roof(kernel, workload, prof_entry, bound, work, note, traffic=(None, None)):
    """One per-kernel roofline entry: `work` = compulsory bytes (bound 'hbm') or flop (bound 'mfma') per launch;
    `traffic` = (HBM bytes per launch from the committed counter profile of the same workload, its source)."""
    cnt, ms = prof_entry
    avg_s = ms / cnt * 1e-3
    if bound == "hbm":
        achieved, peak, unit = work / avg_s / 1e9, HBM_PEAK_GBS, "GB/s"
    else:
        achieved, peak, unit = work / avg_s / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s"
    out = {"kernel": kernel, "workload": workload, "bound": bound, "achieved": achieved, "peak": peak, "unit": unit,
           "frac": achieved / peak, "per_launch": work, "avg_launch_us": avg_s * 1e6, "launches_timed": cnt,
           "note": note}
    if traffic[0]:
        out["traffic"], out["traffic_source"] = traffic[0], traffic[1]
    return out


def other_configs(device):
    """Quick single-GPU numbers for the remaining BASELINE.json configs (not the headline metric; parity for these
    shapes is covered by tests/), and a roofline entry for the dominant kernel of each (HIP-event timed, like the
    headline kernel).  cfg4 / cfg5 are run at the per-GPU shard of their 8-GPU batch."""
    res, roofs = {}, []
    g = torch.Generator(device=device)
    g.manual_seed(77)
    # cfg2: batch 64 low-rank + diag, end-to-end solve (preconditioner build + CG)
    Cm = torch.randn(64, N, R, generator=g, device=device) / (R ** 0.5)
    d = torch.rand(64, N, generator=g, device=device) + 0.5
    rhs = torch.randn(64, N, 1, generator=g, device=device)
    from linear_operator_amd import settings as lo_settings
    from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
    from linear_operator_amd.operators.added_diag_linear_operator import clear_preconditioner_memo

    A2 = AddedDiagLinearOperator(LowRankRootLinearOperator(Cm), DiagLinearOperator(d))

    def cfg2_solve():  # the public path, nothing memoised: one resident launch (lo_solve_fused_f32)
        clear_preconditioner_memo()
        return A2.solve(rhs)

    with lo_settings.cg_tolerance(TOL):
        t, x2 = _time(cfg2_solve, 5)
    res["cfg2_B64_solve_end_to_end"] = {"ms": t * 1e3, "solves_per_s": 64 / t, "path": "A.solve(rhs), operator API",
                                        "solve_rel_err": woodbury_fp64_rel_err(Cm, d, rhs, x2)}
    with lo_settings.cg_tolerance(TOL):
        prof = _profiled(cfg2_solve)
    clear_preconditioner_memo()
    if "solve_fused" in prof:
        roofs.append(_roof("k_solve_fused<32,8,false>", "cfg2: 64 members, pivoted Cholesky + root form + 11 CG iterations "
                           "in one launch", prof["solve_fused"], "hbm", 4 * 64 * (N * (R + 3) + 3 * R * R + N),
                           "compulsory bytes per launch: C, d, rhs in; x, 1/d, F, EF, E out.  64 members = one round of "
                           "the 64 resident groups: the launch time is the latency of ONE member's chain of 15 pivot "
                           "exchanges and 12 CG all-reduces"))
    desc = K.lowrank_diag_descriptor(Cm, d)
    # cfg3: batch 512, 16 probes + 1 rhs, CG with tridiagonals + SLQ logdet (preconditioner build included)
    Cm = torch.randn(B_PER_GPU, N, R, generator=g, device=device) / (R ** 0.5)
    d = torch.rand(B_PER_GPU, N, generator=g, device=device) + 0.5
    full = torch.randn(B_PER_GPU, N, 17, generator=g, device=device)
    full[..., :16] /= full[..., :16].norm(dim=-2, keepdim=True)
    desc = K.lowrank_diag_descriptor(Cm, d)

    def iql():
        # (root form + fp64 Gram matrices, no generic Q: what AddedDiagLinearOperator._init_cache builds since round 5 --
        #  Q is built on demand by the engines that take it, never on this path)
        pre = build_precond(desc, d, need_q=False)
        r = K.cg_solve(desc, full, precond=pre, n_tridiag=16, tolerance=TOL)
        _, _, ld = K.tridiag_eigh_slq(r.t_mat, N)
        return r, ld + pre.logdet

    t, (r, _) = _time(iql, 2)
    res["cfg3_B512_inv_quad_logdet"] = {"ms": t * 1e3, "member_solve_logdets_per_s": B_PER_GPU / t,
                                        "iterations": r.iterations,
                                        "member_matvec_columns_per_s": B_PER_GPU * 17 * r.matvecs / t}
    prof = _profiled(iql)
    it3 = r.iterations
    if "cg_lockstep" in prof:  # 16 probe columns in lockstep on the matrix cores
        flop = 2.0 * B_PER_GPU * N * (2 * R + 2 * RANK_K) * 16 * it3
        roofs.append(_roof("k_cg_lockstep<32,pre>", "cfg3: 16 probe columns x 21 iterations, 512 members", prof["cg_lockstep"],
                           "mfma", flop,
                           "fp32 matrix cores (v_mfma_f32_16x16x4_f32): 2 N (2R + 2k) flop per member, column and "
                           "iteration (k = 15; the kernel multiplies with Q zero-padded to 16); compulsory HBM bytes per "
                           f"launch {4 * B_PER_GPU * N * (R + RANK_K + 2 + 6 * 16) / 1e9:.2f} GB (operator once, rhs in, "
                           "x / r / p / z and the scaled result out)", _committed_traffic("traffic_lockstep.json")))
    if "cg_onchip" in prof:  # the 17th column (inv_quad right-hand side) on the serial-column resident kernel
        roofs.append(_roof("k_cg_onchip5<32,8,MODE 1>", "cfg3: 17th column x 21 iterations, 512 members", prof["cg_onchip"],
                           "hbm", 4 * B_PER_GPU * N * (R + RANK_K + 2 + 6),
                           "compulsory bytes per launch (operator once + one column's vectors); latency-bound by the "
                           "per-iteration group all-reduces"))
    if "pc_onchip" in prof:
        roofs.append(_roof("k_pc_onchip4<32,8>", "rank-15 pivoted Cholesky of 512 x (8192 x 32) roots", prof["pc_onchip"],
                           "hbm", 4 * B_PER_GPU * (N * R + RANK_K * N + N),
                           "compulsory bytes per launch: C once in, the 15 rows of L and the permutation out; bound by "
                           "one group exchange per pivot"))
    # explicit Lanczos on the same operator: 16 probe vectors, 20 steps, full re-orthogonalisation
    V = torch.randn(B_PER_GPU, N, 16, generator=g, device=device)
    t, (q_mat, _) = _time(lambda: K.lanczos_tridiag(desc, V, 20), 1)
    steps = q_mat.shape[-1]
    lz_bytes = sum(4 * B_PER_GPU * ((N * R + N + 2 * N * 16) + 2 * (j + 1) * N * 16) for j in range(steps))
    res["cfg3_B512_lanczos_P16_k20"] = {"ms": t * 1e3, "member_probe_steps_per_s": B_PER_GPU * 16 * steps / t}
    roofs.append({"kernel": "lanczos_tridiag (whole call: matvec + Gram-Schmidt kernels of every step)",
                  "workload": "512 members, 16 probes, 20 steps", "bound": "hbm",
                  "achieved": lz_bytes / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": lz_bytes / t / 1e9 / HBM_PEAK_GBS,
                  "per_launch": lz_bytes, "avg_launch_us": t * 1e6, "launches_timed": 1,
                  "note": "SURVEY 8(d) model per step j: matvec 4(NR + N + 2NP) + re-orthogonalisation 4 * 2 (j+1) N P "
                          "(Q_{<=j} read once for the coefficients, once for the correction), wall time of the call"})
    # the same operator through the host API, forward + backward (one GP marginal-likelihood gradient step): probes
    # drawn from the preconditioner, resident CG with tridiagonals, SLQ, then the reference's backward formulas
    from linear_operator_amd import settings as lo_settings
    from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
    from linear_operator_amd.operators.added_diag_linear_operator import clear_preconditioner_memo

    Cg, dg = Cm.clone().requires_grad_(True), d.clone().requires_grad_(True)
    y = full[..., 16:].contiguous()

    def train_step():
        # a real training step changes the leaves every step: drop the memoised factorisation so that the pivoted
        # Cholesky and the preconditioner build are part of every timed step (the backward solve of the SAME step
        # still reuses the forward's preconditioner, as the reference's cached operator does)
        clear_preconditioner_memo()
        Cg.grad = dg.grad = None
        A = AddedDiagLinearOperator(LowRankRootLinearOperator(Cg), DiagLinearOperator(dg))
        iq, ld = A.inv_quad_logdet(y, logdet=True)
        (iq.sum() + ld.sum()).backward()
        return iq

    with lo_settings.cg_tolerance(TOL), lo_settings.num_trace_samples(16):
        # (one untimed + two timed steps, as before: the cyclic collector is off in this process and every step's autograd
        # graph holds several GB until it is collected -- more repetitions would time the allocator, not the step)
        # Every step timed on its own, the median reported (with the per-step list): a mean over two steps right behind
        # the Lanczos extra measured that extra's memory coming back through the caching allocator (7.9 / 22.8 ms for the
        # same build in two runs; tools/mb_train_step.py: 7.3 - 7.5 ms per step in steady state).
        # Round 6: FOUR untimed steps -- the first three steps behind the Lanczos extra still run at 23 ms while its memory
        # comes back through the caching allocator (profiles/r06/bench.json of the first profile run: 23.6 / 23.6 / 22.6 /
        # 7.7 / 7.6 ms; the median of five was 22.6).
        for _ in range(4):
            train_step()
            torch.cuda.synchronize()
        laps = []
        for _ in range(5):
            ta = time.perf_counter()
            out = train_step()
            torch.cuda.synchronize()
            laps.append(time.perf_counter() - ta)
            del out
        t = sorted(laps)[len(laps) // 2]
    res["cfg3_B512_inv_quad_logdet_forward_backward_host_api"] = {"ms": t * 1e3, "member_steps_per_s": B_PER_GPU / t,
                                                                  "steps_ms": [round(x * 1e3, 3) for x in laps],
                                                                  "preconditioner": "rebuilt every step"}
    del Cm, d, full, desc, V, q_mat, Cg, dg, y
    # cfg4 shard: 128 of 1024 Kronecker members (256 (x) 256 + 1e-2 I), CG to tolerance 1e-3
    n = 256
    X1 = torch.randn(128, n, n, generator=g, device=device) / 16
    X2 = torch.randn(128, n, n, generator=g, device=device) / 16
    K1 = X1 @ X1.mT + 0.1 * torch.eye(n, device=device)
    K2 = X2 @ X2.mT + 0.1 * torch.eye(n, device=device)
    sig = torch.full((128,), 1e-2, device=device)
    rhs = torch.randn(128, n * n, 1, generator=g, device=device)
    desc = K.kron_diag_descriptor(K1, K2, sig, const_diag=True)

    kd = desc.without_diag()  # the preconditioner factors the Kronecker part (added_diag_linear_operator.py:125)

    def kron():
        L, perm = K.pivoted_cholesky(kd, RANK_K, contiguous=False)  # [B, m, N] rows read in place by the build
        # (perm + the operator: the build adds the Kronecker root form, what AddedDiagLinearOperator._init_cache passes)
        return K.cg_solve(desc, rhs, precond=K.precond_build(L, sig, True, perm=perm, kron=desc), tolerance=1e-3)

    t, r = _time(kron, 1)
    res["cfg4_shard_B128_kron_solve"] = {"ms": t * 1e3, "solves_per_s": 128 / t, "iterations": r.iterations}
    prof = _profiled(kron)
    if "kron_fused" in prof:
        roofs.append(_roof("k_kron_fused<8,4>", "cfg4 shard: 128 members, BOTH 256^3 GEMMs of a Kronecker matvec in one "
                           "launch", prof["kron_fused"], "mfma", 4.0 * 128 * n * n * n,
                           "fp32 matrix cores (v_mfma_f32_32x32x2_f32), 2 x 2 n^3 flop per member; the intermediate "
                           "V K2^T stays in the accumulators and is the second GEMM's B operand as it lies (DESIGN 4.8); "
                           "compulsory HBM bytes per launch 134 MB (K1, K2, v in, y out)",
                           _committed_traffic("traffic_cfg45.json", "kron_fused")))
    if "precond_fused_kron" in prof:
        roofs.append(_roof("k_precond_fused_kron<32,4,8>", "cfg4 shard: 128 members x 65536 rows, one CG iteration's "
                           "Woodbury apply in Kronecker root form + r / x / p updates", prof["precond_fused_kron"], "hbm",
                           4 * 128 * n * n * 7,
                           "4 N 7 bytes per member and iteration: r, Ap, p, x in; r, x, p out -- the rows of the "
                           "preconditioner's tall matrix are formed on the fly from the pivot rows of the two factors "
                           "(DESIGN 4.7); bound by one group hand-off per member at four rounds of 32 groups",
                           _committed_traffic("traffic_cfg45.json", "precond_fused_kron")))
    if "kron_gemm_mfma" in prof:
        roofs.append(_roof("k_kron_nt_mfma", "cfg4 shard: 128 members, one of the two 256^3 GEMMs of a Kronecker matvec",
                           prof["kron_gemm_mfma"], "mfma", 2.0 * 128 * n * n * n,
                           "fp32 matrix cores (v_mfma_f32_32x32x2_f32), 2 n^3 flop per member and GEMM; with K = 256 "
                           "the GEMM sits at the ridge (operands + result 100 - 130 MB per launch ~ its MFMA time): "
                           "tools/mb_kron_rate.py, DESIGN 4.6", _committed_traffic("traffic_cfg45.json", "kron_gemm_mfma")))
    if "precond_fused" in prof:
        roofs.append(_roof("k_precond_fused<64,4>", "cfg4 shard: 128 members x 65536 rows, one CG iteration's Woodbury "
                           "apply + r / x / p updates", prof["precond_fused"], "hbm", 4 * 128 * n * n * (16 + 7),
                           "4 N (16 + 7) bytes per member and iteration: Q once; r, Ap, p, x in; r, x, p out (constant "
                           "diagonal)", _committed_traffic("traffic_cfg45.json", "precond_fused")))
    del X1, X2, K1, K2, desc
    # cfg5 shard: 4 of the 32 dense 16384^2 members a GPU owns, 17 columns, CG with tridiagonals
    Nd = 16384
    X = torch.randn(4, Nd, Nd, generator=g, device=device) / 128
    Kd = X @ X.mT
    del X
    d = torch.rand(4, Nd, generator=g, device=device) + 0.5
    full = torch.randn(4, Nd, 17, generator=g, device=device)
    desc = K.dense_diag_descriptor(Kd, d)

    def dense():
        L, _ = K.pivoted_cholesky(desc, RANK_K, contiguous=False)  # [B, m, N] rows read in place by the build
        return K.cg_solve(desc, full, precond=K.precond_build(L, d, False), n_tridiag=16, tolerance=TOL)

    t, r = _time(dense, 1)
    mv_bytes = 4 * 4 * (Nd * Nd + Nd + 2 * Nd * 17)
    res["cfg5_shard_B4_dense_cg"] = {"ms": t * 1e3, "iterations": r.iterations,
                                     "matvec_algorithmic_GBs": mv_bytes * r.matvecs / t / 1e9}
    prof = _profiled(dense)
    if "dense_mv_mfma" in prof:
        roofs.append(_roof("k_dense_mv_mfma16<true,1>", "cfg5 shard: 4 members of 16384^2, 17 columns, one matvec",
                           prof["dense_mv_mfma"], "hbm", mv_bytes,
                           "4 (N^2 + N + 2 N c) bytes per member: K streamed once for all 17 columns (the committed "
                           "counter profile is of 8 members: twice the bytes)"))
    if "cg_step_cols" in prof:
        roofs.append(_roof("k_cg_step_cols<64,2,true,1>", "cfg5 shard: 4 members of 16384 rows, 17 columns: everything of a CG "
                           "iteration behind the product (alpha, r / x, Q-form preconditioner, beta, p, control step with "
                           "tridiagonals) in one launch", prof["cg_step_cols"], "hbm", 4 * 4 * Nd * (7 * 17 + 16 + 1),
                           "4 N (7 c + 16 + 1) bytes per member: r, Ap, p, x in; r, x, p out; Q and 1/d once.  One member per "
                           "group of 64 workgroups, one exchange: a chain of latencies (loads 10 us, exchange 12 us), not a "
                           "stream -- it replaced three launches of 145 us (the committed counter profile, traffic_cfg45.json, is of 8 "
                           "members)"))
    return res, roofs


def strong_scaling(args, device, dist, rank, world):
    """`--workload cfg4|cfg5`: BASELINE's sharded configs at their FULL batch (1024 Kronecker members / 256 dense
    16384^2 members), split over the ranks (strong scaling).  Every rank constructs only its own members on its own
    device (distributed.build_local_shard semantics); when even the shard exceeds the memory budget (cfg5 on one
    GPU: 256 GiB) the shard is processed in resident chunks and the chunks reuse one chunk's synthetic data.
    A step = preconditioner (pivoted Cholesky + build) + CG (+ tridiagonals and SLQ for cfg5) over the whole batch."""
    from linear_operator_amd.distributed import shard_bounds

    total = 1024 if args.workload == "cfg4" else 256
    lo, hi = shard_bounds(total, rank, world)
    mine = hi - lo
    g = torch.Generator(device=device)
    g.manual_seed(4000 + rank)
    if args.workload == "cfg4":
        n = 256
        chunk = mine
        X1 = torch.randn(chunk, n, n, generator=g, device=device) / 16
        X2 = torch.randn(chunk, n, n, generator=g, device=device) / 16
        K1 = X1 @ X1.mT + 0.1 * torch.eye(n, device=device)
        K2 = X2 @ X2.mT + 0.1 * torch.eye(n, device=device)
        del X1, X2
        sig = torch.full((chunk,), 1e-2, device=device)
        rhs = torch.randn(chunk, n * n, 1, generator=g, device=device)
        desc = K.kron_diag_descriptor(K1, K2, sig, const_diag=True)

        kd = desc.without_diag()

        def solve_chunk():
            L, perm = K.pivoted_cholesky(kd, RANK_K, contiguous=False)
            return K.cg_solve(desc, rhs, precond=K.precond_build(L, sig, True, perm=perm, kron=desc), tolerance=1e-3)

        cols, what = 1, "KroneckerProduct(256x256, 256x256) + 1e-2 I, N = 65536, 1 rhs column, tolerance 1e-3"
    else:
        Nd = 16384
        chunk = min(mine, max(1, int(args.chunk_members)))
        Y = torch.randn(chunk, Nd, 256, generator=g, device=device) / 16
        Kd = Y @ Y.mT
        del Y
        d = torch.rand(chunk, Nd, generator=g, device=device) + 0.5
        full = torch.randn(chunk, Nd, 17, generator=g, device=device)
        full[..., :16] /= full[..., :16].norm(dim=-2, keepdim=True)
        desc = K.dense_diag_descriptor(Kd, d)

        def solve_chunk():
            L, _ = K.pivoted_cholesky(desc, RANK_K, contiguous=False)
            pre = K.precond_build(L, d, False)
            r = K.cg_solve(desc, full, precond=pre, n_tridiag=16, tolerance=TOL)
            K.tridiag_eigh_slq(r.t_mat, Nd)
            return r

        cols, what = 17, "AddedDiag(Dense 16384^2 (rank-256 PSD), Diag), 16 probes + 1 rhs, CG + SLQ logdet"
    nchunks = (mine + chunk - 1) // chunk

    def step():
        r = None
        for _ in range(nchunks):
            r = solve_chunk()
        return r

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    res = None
    for _ in range(args.warmup):
        res = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return None
    return {
        "metric": "cg_member_matvecs_per_sec", "value": total * res.matvecs * args.steps / elapsed,
        "unit": "member-matvecs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE {args.workload} at its full batch of {total} members: {what}",
                   "total_members": total, "members_per_rank": mine, "resident_chunk": chunk,
                   "chunks_per_step": nchunks, "rhs_columns": cols, "iterations": res.iterations,
                   "note": ("every rank builds only its own members; " +
                            ("chunks beyond the resident one reuse its synthetic data (the shard exceeds the memory "
                             "budget)" if nchunks > 1 else "the whole shard is resident")),
                   "sharding": f"batch x{world}, no collective inside the solve"},
    }


def self_launch(n, argv):
    """Re-exec this script under torch.distributed.run with n ranks on this node (rendezvous on 127.0.0.1, a free
    port); rank 0's JSON line goes to our stdout.  Returns the launcher's exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def _step_stats(laps):
    """min / median / p95 / max of per-step host times in ms."""
    a = sorted(laps)
    n = len(a)
    return {"steps": n, "min_ms": a[0] * 1e3, "median_ms": a[n // 2] * 1e3, "p95_ms": a[min(n - 1, int(0.95 * n))] * 1e3,
            "max_ms": a[-1] * 1e3, "mean_ms": sum(a) / n * 1e3}


def _gather_rank_ms(dist, elapsed_ms, device):
    """Per-rank step times (ms) on every rank."""
    t = torch.tensor([elapsed_ms], device=device, dtype=torch.float64)
    allt = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(allt, t)
    return [float(x.item()) for x in allt]


def stub_run(args, rank, world):
    """LO_BENCH_STUB=1: the launcher, the rank plumbing, barrier + max-over-ranks timing and the JSON contract with a
    stubbed solver on the gloo backend -- runs without a GPU.  Never a measurement: metric and data say `stub`."""
    import torch.distributed as dist

    if world > 1 or os.environ.get("LO_BENCH_FORCE_DIST"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("gloo")
        assert dist.get_world_size() == args.gpus
    else:
        dist = None
    dev = torch.device("cpu")
    x = torch.full((4, 8, 1), float(rank))
    buf = torch.empty(world * 4, 8, 1)

    def step():
        time.sleep(0.002 * (1 + rank))  # uneven ranks: the MAX must win
        if dist is not None:
            dist.all_gather_into_tensor(buf, x)

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = [elapsed / args.steps * 1e3]
    if dist is not None:
        rank_ms = _gather_rank_ms(dist, rank_ms[0], dev)
        ok = bool(torch.equal(buf[:, 0, 0].reshape(world, 4)[:, 0], torch.arange(world, dtype=torch.float32)))
        dist.barrier()
        dist.destroy_process_group()
    else:
        ok = True
    ms = max(rank_ms)
    return {"metric": "stub", "value": world * 4 * 11 / (ms * 1e-3), "unit": "member-matvecs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "stub", "config": {"workload": "stub (launcher check, no GPU)"},
            "rccl_ranks": world, "per_rank_ms": rank_ms, "gather_ok": ok}


def main():
    # stdout carries exactly ONE JSON line: everything else that libraries print to fd 1 (RCCL prints a version banner
    # on process-group setup) is sent to stderr for the whole run; the JSON goes to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the quick numbers for the other BASELINE configs")
    ap.add_argument("--workload", choices=["headline", "cfg4", "cfg5"], default="headline",
                    help="headline (default, weak scaling, the driver's contract) or BASELINE cfg4 / cfg5 at their "
                         "full batch split over the ranks (strong scaling)")
    ap.add_argument("--min-seconds", type=float, default=3.0,
                    help="after the timed K steps: keep stepping (each step timed on its own) for about this much wall "
                         "time -- per-step min / median / max for the JSON line, and a GPU that an external "
                         "utilisation sampler sees busy.  0 = off.  `value` / `ms_per_step` come from the K steps only")
    ap.add_argument("--gather", choices=["step", "end"], default="end",
                    help="N > 1: `end` (default) = ONE all-gather of the solutions after the K steps, inside the timed "
                         "region (north_star: `an RCCL all-gather over xGMI at the end`); `step` = one all-gather per "
                         "solve, overlapped with the next solve (the demanding reading: 16 MB per rank every 0.25 ms).  "
                         "The other mode is timed over the same K steps as well and reported under `gather_modes`")
    ap.add_argument("--inject-timeouts", type=int, default=0,
                    help="treat the first n resident CG launches of the soak as timed out (exercises the cool-down / "
                         "re-arm gate of the resident kernels; the JSON line reports what the gate did)")
    ap.add_argument("--chunk-members", type=int, default=128,
                    help="cfg5: members resident at a time per rank (1 GiB each)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU under torch.distributed.run,
        # the driver's own convention) instead of dying -- the scaling run must not be lost to a launch convention
        os.dup2(real_stdout, 1)
        os.close(real_stdout)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if os.environ.get("LO_BENCH_STUB"):  # launcher / JSON-contract check without a GPU (tests/test_distributed_cpu.py)
        out = stub_run(args, rank, world)
        if rank == 0:
            os.write(real_stdout, (json.dumps(out) + "\n").encode())
        os.close(real_stdout)
        return
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: no HIP device {local_rank} (visible devices: "
                         f"{torch.cuda.device_count() if torch.cuda.is_available() else 0}); bench.py has no CPU path")
    dist = None
    use_dist = world > 1 or bool(os.environ.get("LO_BENCH_FORCE_DIST"))  # the env var exercises the RCCL path at N=1
    if use_dist:
        import torch.distributed as dist  # noqa: PLW0621

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        # at most 32 RCCL channels (= workgroups): the resident kernels leave 32 CUs' worth of slots (LO_OC_RESERVE_CUS
        # below) and a 16 MB per-rank all-gather does not need more
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "32")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    _hip.load()
    _gc_off()

    if args.workload != "headline":
        out = strong_scaling(args, device, dist if use_dist else None, rank, world)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        sys.stdout.flush()
        if rank == 0:
            os.write(real_stdout, (json.dumps(out) + "\n").encode())
        os.close(real_stdout)
        return

    reserve_default = None
    if use_dist:
        # (--gather step) The all-gather of solve k runs on RCCL's stream while solve k + 1 computes.  The resident CG kernel fills every
        # CU with two workgroups; a workgroup RCCL's kernel displaces stalls its whole group until the gather ends.
        # Leaving 32 CUs' worth of slots unused (liblo_amd reads the variable at every launch) gives RCCL room.
        reserve_default = os.environ.get("LO_OC_RESERVE_CUS")  # (an explicit setting of the caller wins in both modes)
    # The quick numbers of the other BASELINE configs run FIRST (single GPU): several seconds of the library's own
    # kernels, after which the device sits at its steady clocks -- right after start-up the same solve is ~7 % slower
    # and speeds up over its first ~40 launches (tools/mb_step_ramp.py: 0.400 -> 0.372 ms per step), a ramp that would
    # otherwise fall into the W warm-up + K timed steps.  The timed region itself is exactly the contract's.
    extras, extras_error = None, None
    if world == 1 and rank == 0 and not args.no_extras:
        try:
            extras = other_configs(device)
        except Exception as e:  # noqa: BLE001 -- the headline line must not be lost to an extra
            extras_error = repr(e)
        torch.cuda.empty_cache()
    Cm, d, rhs = make_problem(device, 1234 + rank)
    desc = K.lowrank_diag_descriptor(Cm, d)
    pre = build_precond(desc, d)
    # The single collective of the path (north_star): one all-gather of the solutions per solve.  It is issued
    # asynchronously (RCCL's own stream, after the solve's kernels) and double-buffered, so the gather of solve k
    # travels over xGMI while solve k + 1 computes; every gather is complete before the timed region ends.
    gather_bufs = [torch.empty(world * B_PER_GPU, N, C_COLS, device=device) for _ in range(2)] if use_dist else None
    pending = []  # [(work handle, tensor being gathered)]
    nstep = [0]

    gather_mode = [None]

    def set_gather_mode(m):
        """`step`: RCCL's all-gather kernel runs beside the solves -- leave it 32 CUs' worth of workgroup slots."""
        gather_mode[0] = m
        if use_dist and reserve_default is None:
            os.environ["LO_OC_RESERVE_CUS"] = "32" if m == "step" else "0"

    set_gather_mode(args.gather)
    engines_seen = {}

    def step():
        res = K.cg_solve(desc, rhs, precond=pre, tolerance=TOL)
        if use_dist and gather_mode[0] == "step":
            while pending:  # the previous gather has had a whole solve to finish
                pending.pop(0)[0].wait()
            buf = gather_bufs[nstep[0] % 2]
            nstep[0] += 1
            pending.append((dist.all_gather_into_tensor(buf, res.x, async_op=True), res.x))
        return res

    def fence(last=None):
        """Both ends of a timed region.  `last` (gather mode `end`): the solution of the region's final solve -- its
        ONE all-gather belongs to the region."""
        if use_dist:
            while pending:
                pending.pop(0)[0].wait()
            if last is not None and gather_mode[0] == "end":
                dist.all_gather_into_tensor(gather_bufs[0], last.x)
            dist.barrier()
        torch.cuda.synchronize(device)

    def engine_now():
        """Which engine the last solve of this rank ran (lo_cg_last_executed)."""
        e = K.cg_last_executed()
        if not e["resident"]:
            return "streaming"
        if e["rspace"] == "resident" and e.get("rspace_diag"):
            return "resident R-space, diagonal form (k_cg_rspace3)"
        return {"resident": "resident R-space (k_cg_rspace)", "cols": "R-space columns (k_rs_part/iter/apply)",
                "none": "resident three-pass (k_cg_onchip5)"}[e["rspace"]]

    # ---- this box's HBM ceilings (reported beside the spec peak), measured first: 1 GiB arrays ----
    triad_gbs = copy_gbs = None
    if rank == 0:
        try:
            triad_gbs = round(_hip.hbm_stream_gbs(device, "triad"), 1)
            copy_gbs = round(_hip.hbm_stream_gbs(device, "copy"), 1)
        except Exception:  # noqa: BLE001
            pass
        torch.cuda.empty_cache()
    # ---- validation BEFORE the timed region: one solve checked against the fp64 Woodbury closed form of the same systems
    # (the exact solution; torch fp64 library ops as the CHECKER).  The kernels are bitwise reproducible, so this is the
    # very result every timed step computes (asserted after the timed region).
    res_check = K.cg_solve(desc, rhs, precond=pre, tolerance=TOL)
    solve_rel_err = woodbury_fp64_rel_err(Cm, d, rhs, res_check.x)
    x_check = res_check.x.clone()
    res = None
    # soak FIRST: the same step again and again for >= --min-seconds, every step timed on its own (host clock around a
    # synchronous solve).  It doubles as the ramp: the first ~40 solves after ANY pause of the device (the validation
    # above reads results back on the host) are 3 - 7 % slower than the steady state (tools/mb_step_ramp.py), and a W + K
    # window of 10 ms right after such a pause measured the ramp, not the engine (0.404 vs 0.376 ms on the same box,
    # profiles/r04/README.md).  The step count comes from a fenced pilot of five steps, maximum over the ranks, so that
    # every rank runs the same number of steps (and all-gathers).
    soak = None
    if args.min_seconds > 0:
        for _ in range(3):
            res = step()
        fence()
        tp = time.perf_counter()
        for _ in range(5):
            res = step()
        fence()
        pilot = (time.perf_counter() - tp) / 5
        if use_dist:
            pilot = max(_gather_rank_ms(dist, pilot * 1e3, device)) * 1e-3
        n_soak = max(1, min(200000, int(args.min_seconds / pilot)))
        laps = [0.0] * n_soak
        if args.inject_timeouts > 0:
            K.inject_resident_timeouts(args.inject_timeouts)
        for i in range(n_soak):
            ta = time.perf_counter()
            res = step()
            laps[i] = time.perf_counter() - ta
            if args.inject_timeouts > 0:
                en = engine_now()
                engines_seen[en] = engines_seen.get(en, 0) + 1
        fence()
        soak = _step_stats(laps)
        soak["seconds"] = round(sum(laps), 3)
        soak["position"] = "before the warm-up and the timed steps"
    # ---- the contract's region: W untimed warm-up steps, then EXACTLY K steps between two fences (barrier + synchronize)
    for _ in range(args.warmup):
        res = step()
    fence()
    stamps = [0.0] * (args.steps + 1)
    t0 = stamps[0] = time.perf_counter()
    for i in range(args.steps):
        res = step()
        stamps[i + 1] = time.perf_counter()  # (linear_cg returns when the solve's status block has arrived)
    fence(res)
    elapsed = time.perf_counter() - t0
    engine_timed = engine_now()
    gate = K.resident_status()
    step_stats = _step_stats([stamps[i + 1] - stamps[i] for i in range(args.steps)])
    per_rank_ms = [elapsed / args.steps * 1e3]
    allgather_ms = None
    if use_dist:
        per_rank_ms = _gather_rank_ms(dist, per_rank_ms[0], device)
        elapsed = max(per_rank_ms) * 1e-3 * args.steps  # the MAX over the ranks is the job's time
        # the collective on its own (blocking, nothing to hide behind): what one all-gather of the solutions costs
        torch.cuda.synchronize(device)
        dist.barrier()
        tg = time.perf_counter()
        for i in range(5):
            dist.all_gather_into_tensor(gather_bufs[i % 2], res.x)
        torch.cuda.synchronize(device)
        allgather_ms = (time.perf_counter() - tg) / 5 * 1e3
    matvecs_per_solve = res.matvecs
    total_members = world * B_PER_GPU
    value = total_members * matvecs_per_solve * args.steps / elapsed

    def timed_region(n_steps):
        """W warm-up + n_steps timed steps between two fences, maximum over the ranks (ms per step)."""
        r_ = None
        for _ in range(args.warmup):
            r_ = step()
        fence()
        ta = time.perf_counter()
        for _ in range(n_steps):
            r_ = step()
        fence(r_)
        ms = (time.perf_counter() - ta) / n_steps * 1e3
        return max(_gather_rank_ms(dist, ms, device)) if use_dist else ms

    # ---- the other gather mode over the same K steps (N > 1): both readings of north_star's "all-gather at the end" ----
    gather_modes = None
    if use_dist:
        other = "end" if args.gather == "step" else "step"
        set_gather_mode(other)
        other_ms = timed_region(args.steps)
        set_gather_mode(args.gather)
        gather_modes = {args.gather: elapsed / args.steps * 1e3, other: other_ms, "timed_as_value": args.gather,
                        "note": "`step`: one all-gather of the solutions per solve, overlapped with the next solve; "
                                "`end`: ONE all-gather after the K steps, inside the timed region"}
    # ---- the same K steps on the resident kernel that performs the 11 products on the rows (k_cg_onchip5, the engine
    # of rounds 2 - 4 and still the repeat-with-state / global-stop-rule engine): the R-space pass switched off ----
    def all_ranks(flag):
        """The comparison regions below contain barriers: every rank has to enter them or none (a rank whose timed solves
        fell back to another engine would otherwise leave the others waiting)."""
        return (min(_gather_rank_ms(dist, 1.0 if flag else 0.0, device)) > 0.5) if use_dist else bool(flag)

    matvec_engine = None
    if all_ranks(engine_timed.startswith("resident R-space")):
        os.environ["LO_OC_NO_RSPACE"] = "1"
        try:
            ms3 = timed_region(args.steps)
            eng3 = engine_now()
        finally:
            del os.environ["LO_OC_NO_RSPACE"]
        matvec_engine = {"engine": eng3, "ms_per_step": ms3, "value": total_members * matvecs_per_solve / (ms3 * 1e-3),
                         "note": "LO_OC_NO_RSPACE=1: the 11 operator applications carried out on the rows of C in VGPRs "
                                 "(three passes over the registers per iteration); same systems, same stop rule"}
    # ---- the same K steps on the DENSE R-space iteration (four R x R products per iteration; the form a fresh cache
    # carries before its second solve) and the cost of the diagonal form this rank's cache was given ----
    rspace_dense_engine = None
    if all_ranks(engine_timed.startswith("resident R-space, diagonal")):
        os.environ["LO_RS_NO_DIAG"] = "1"
        try:
            ms4 = timed_region(args.steps)
            eng4 = engine_now()
        finally:
            del os.environ["LO_RS_NO_DIAG"]
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(5):
            pre.RSD = None
            pre.rsd_refused = None
            pre.ensure_eigform()
        torch.cuda.synchronize(device)
        rspace_dense_engine = {"engine": eng4, "ms_per_step": ms4,
                               "value": total_members * matvecs_per_solve / (ms4 * 1e-3),
                               "eigform_build_ms": (time.perf_counter() - t0) / 5 * 1e3,
                               "note": "LO_RS_NO_DIAG=1: k_cg_rspace<..,false>, the iteration with the four R x R products; "
                                       "eigform_build_ms = lo_precond_eigform_f32 for this rank's members (two Jacobi "
                                       "eigendecompositions each), paid ONCE per preconditioner cache when it serves its "
                                       "25th single-column solve (kernels._eigform_due: when the solves served have paid for it) -- "
                                       "outside the timed region, as the pivoted Cholesky and the root form are"}
    # ---- per-rank engine and gate of the resident kernels: a line measured on the streaming engine says so ----
    engines = [engine_timed]
    gates = [gate]
    if use_dist:
        engines = [None] * world
        gates = [None] * world
        dist.all_gather_object(engines, engine_timed)
        dist.all_gather_object(gates, gate)

    # ---- end-to-end solves/s (pivoted Cholesky + preconditioner build + CG), local rank ----
    # Through the PUBLIC path: A.solve(rhs) of AddedDiagLinearOperator(LowRankRootLinearOperator(C), DiagLinearOperator(d))
    # -> Solve Function -> _solve_preconditioner -> utils.linear_cg, nothing memoised (the memo is cleared before every
    # solve: a new operator every time, as in a training loop).  This is the one-launch kernel lo_solve_fused_f32.
    from linear_operator_amd import settings as lo_settings
    from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator
    from linear_operator_amd.operators.added_diag_linear_operator import clear_preconditioner_memo

    A_op = AddedDiagLinearOperator(LowRankRootLinearOperator(Cm), DiagLinearOperator(d))

    def e2e_api():
        clear_preconditioner_memo()
        return A_op.solve(rhs)

    def e2e_three_launch():  # the same work as three resident launches (round 2's path), kernel-level calls
        p2 = build_precond(desc, d, need_q=False)
        return K.cg_solve(desc, rhs, precond=p2, tolerance=TOL)

    with lo_settings.cg_tolerance(TOL):
        for _ in range(2):
            x_e2e = e2e_api()
        fence()
        reps = 5
        t1 = time.perf_counter()
        for _ in range(reps):
            x_e2e = e2e_api()
        fence()
        e2e = (time.perf_counter() - t1) / reps
    e2e_three_launch()
    fence()
    t1 = time.perf_counter()
    for _ in range(3):
        e2e_three_launch()
    fence()
    e2e3 = (time.perf_counter() - t1) / 3
    clear_preconditioner_memo()
    if use_dist:
        t = torch.tensor([e2e], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        e2e = float(t.item())

    # ---- accuracy of the timed solve: max_b ||x - x*|| / ||x*|| against the fp64 Woodbury closed form (the exact
    # solution of the same systems; torch fp64 library ops as the CHECKER, outside every timed region) ----
    # The cache of the timed solves carries the diagonal form of the R-space iteration from its 25th solve on,
    # the validation solve in front of the soak ran the dense form: the result of the LAST TIMED step is validated itself
    # (outside the timed region), and one more solve on the same engine has to reproduce it bit for bit.
    solve_rel_err_first = solve_rel_err
    first_equals_timed = bool(torch.equal(res.x, x_check))
    del x_check
    solve_rel_err = woodbury_fp64_rel_err(Cm, d, rhs, res.x)
    res_again = K.cg_solve(desc, rhs, precond=pre, tolerance=TOL)
    timed_equals_checked = bool(torch.equal(res.x, res_again.x)) and engine_now() == engine_timed
    del res_again
    # ---- BASELINE cfg4 / cfg5 at their full batch over the same ranks (strong scaling), same invocation ----
    strong = {}
    if use_dist and world > 1 and not os.environ.get("LO_BENCH_NO_STRONG"):
        import types

        del pre
        torch.cuda.empty_cache()
        for wl in ("cfg4", "cfg5"):
            a2 = types.SimpleNamespace(workload=wl, steps=2, warmup=1, chunk_members=args.chunk_members)
            line = strong_scaling(a2, device, dist, rank, world)
            if line is not None:
                strong[f"{wl}_strong_scaling"] = {k: line[k] for k in ("value", "unit", "ms_per_step", "scaling", "config")}
            torch.cuda.empty_cache()
        pre = build_precond(desc, d)

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel: live HIP-event timing on the launch stream ----
        _hip.prof_enable(True)
        for _ in range(3):
            K.cg_solve(desc, rhs, precond=pre, tolerance=TOL)
        torch.cuda.synchronize(device)
        prof = _hip.prof_report()
        _hip.prof_enable(False)
        dom = max(prof, key=lambda k: prof[k][1])
        cnt, ms = prof[dom]
        avg_s = ms / cnt * 1e-3
        # ---- the LITERAL batched CG matvec of north_star's 60 % target: one application of the operator to one column
        # through lo_matvec_f32 (= torch.matmul(A, v) of the operator API).  k_lr_mv reads C from HBM ONCE: the rows wait
        # in registers for the group all-reduce of C^T v (csrc/lo_lowrank_mv.hip); live HIP events on the launch stream ----
        hbm_matvec = None
        try:
            def mv_call():
                return K.matvec(desc, rhs)

            t_mv, _ = _time(mv_call, 100)
            prof_mv = _profiled(lambda: [mv_call() for _ in range(50)])
            os.environ["LO_NO_RESIDENT_MV"] = "1"
            try:
                t_mv2, _ = _time(mv_call, 50)
            finally:
                del os.environ["LO_NO_RESIDENT_MV"]
            if "lr_mv" in prof_mv:
                hbm_matvec = _roof(
                    "k_lr_mv<32,8,1>", "lo_matvec_f32 low-rank+diag 512x8192x32, c=1 (one batched CG matvec, C read once)",
                    prof_mv["lr_mv"], "hbm", 4 * B_PER_GPU * (N * R + N + 2 * N * C_COLS),
                    "SURVEY 8(d)'s algorithmic bytes of ONE batched matvec (C, d, v in; y out: 587.2 MB) / the kernel's "
                    "launch time.  The rows of C are read from HBM once and wait in registers (128 per lane) for the "
                    "group all-reduce of t = C^T v; y = C t + d o v comes from the same registers.  north_star's budget: "
                    "<= 122 us (60 % of 8 TB/s)", _committed_traffic("traffic_mv.json"))
                hbm_matvec["wall_us_per_call"] = t_mv * 1e6
                hbm_matvec["member_matvecs_per_sec"] = B_PER_GPU / t_mv
                hbm_matvec["frac_wall"] = 4 * B_PER_GPU * (N * R + N + 2 * N * C_COLS) / t_mv / 1e9 / HBM_PEAK_GBS
                hbm_matvec["two_pass_wall_us_per_call"] = t_mv2 * 1e6
                hbm_matvec["two_pass_note"] = ("LO_NO_RESIDENT_MV=1: k_skinny_tn + k_skinny_nn, C streamed twice (the "
                                               "product of rounds 1 - 5)")
        except Exception as e:  # noqa: BLE001 -- the headline line must not be lost to an extra
            hbm_matvec = {"error": repr(e)}
        # ---- what a solve costs when its preconditioner cache serves n single-column solves: the R-space engine reads
        # E = C^T D^-1 C and C^T C (k_rs_gram64, part of the cache build) and, from the 25th solve on, the diagonal form
        # (k_rs_eigform, built once); the timed region above runs on a cache that has paid both ----
        cache_use = None
        try:
            prof_b = _profiled(lambda: build_precond(desc, d, need_q=False))
            t_build, _ = _time(lambda: build_precond(desc, d, need_q=False), 3)
            gram_ms = (prof_b["rs_gram64"][1] / prof_b["rs_gram64"][0]) if "rs_gram64" in prof_b else None
            cache_use = {"cache_build_ms": t_build * 1e3, "rs_gram64_ms": gram_ms}
        except Exception as e:  # noqa: BLE001
            cache_use = {"error": repr(e)}
        rs_engine = dom == "cg_onchip" and K.cg_last_executed()["rspace"] == "resident"
        rs_diag = rs_engine and K.cg_last_executed().get("rspace_diag", False)
        if rs_diag:
            # k_cg_rspace<..,true>: C, 1/d, the right-hand side in, x out, and four fp64 R x R matrices + R eigenvalues of
            # the diagonal form (TinT | E^+ | TuT | G2 | lam: lo_precond_desc.RSD)
            # (k_cg_rspace3 keeps three of them in LDS -- TinT | TuT | G2 -- and reads the eigenvalues: 3 R^2 + R doubles)
            alg = B_PER_GPU * (4 * N * (R + 3) + 3 * R * R * 8 + 8 * R)
            dom_kernel = "k_cg_rspace3<32,8>"
        elif rs_engine:
            # k_cg_rspace: C, 1/d, the right-hand side in, x out, and the member's six fp64 R x R matrices of the
            # R-space form (E | F E | E F E | G2 | F | E F: lo_precond_desc.RS)
            alg = B_PER_GPU * (4 * N * (R + 3) + 6 * R * R * 8)
            dom_kernel = "k_cg_rspace<32,8,false>"
        else:
            alg = algorithmic_bytes(dom, RANK_K)
            dom_kernel = "k_cg_onchip5<32,8,2>" if dom == "cg_onchip" else dom
        achieved = alg / avg_s / 1e9
        tn, nn = prof.get("skinny_tn_R32"), prof.get("skinny_nn_R32")
        mv_alg = 4 * B_PER_GPU * (N * R + N + 2 * N * C_COLS)  # 1,146,880 B per member
        # SURVEY 8(d)'s per-ITERATION streaming model (matvec + CG state + rank-k preconditioner): what a kernel that
        # re-reads the operator every iteration would have to move.  Reported as an equivalent rate only -- the
        # resident kernel does not move these bytes, so this is NOT a roofline fraction.
        per_iter = 4 * (N * R + N + 2 * N * C_COLS) + 32 * N * C_COLS + 4 * (N * RANK_K + 2 * N * C_COLS)
        equiv = {"bytes_per_member_iteration": per_iter, "iterations": ITERS_FLOOR,
                 "equivalent_GBs": B_PER_GPU * ITERS_FLOOR * per_iter / avg_s / 1e9,
                 "note": "per-iteration streaming model of SURVEY 8(d) divided by the launch time: the rate a "
                         "streaming engine would need to match this kernel; not HBM traffic"}
        if tn and nn:
            mv_s = (tn[1] / tn[0] + nn[1] / nn[0]) * 1e-3
            mv_note = "C^T p + C t + d o p kernel pair = one batched CG matvec (north_star 60% target)"
        else:
            # fused kernel: the matvec cannot be timed on its own; charge it the WHOLE iteration (upper bound)
            mv_s = avg_s / ITERS_FLOOR
            mv_note = ("operator-resident kernel: wall time of one full CG iteration (matvec + preconditioner + "
                       "updates) next to north_star's budget of 122 us for the streamed matvec alone; equivalent rate, "
                       "not HBM traffic -- the literal HBM matvec is `hbm_matvec`")
        kernels = {k: {"launches": v[0], "avg_us": round(v[1] / v[0] * 1e3, 2),
                       "alg_GBs": round(algorithmic_bytes(k, RANK_K) / (v[1] / v[0] * 1e-3) / 1e9, 1)}
                   for k, v in sorted(prof.items())}
        # HBM bytes per launch from rocprofv3 PMC passes: they cannot be collected from inside this process, so the
        # figure comes from the committed profile of this very command (tools/profile_round.sh) and says so.
        traffic, traffic_source = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", PROFILE_DIR, "traffic.json")))
            if tj.get("prof_name") == dom and tj.get("kernel", "").replace(" ", "").startswith(dom_kernel.replace(" ", "").rstrip(">")):
                traffic = tj["traffic_bytes_per_launch"]
                traffic_source = (f"profiles/{PROFILE_DIR}/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                  f"`bench.py --no-extras`, kernel {tj['kernel']}, FETCH x2 (gfx950 correction) + WRITE")
        except Exception:  # noqa: BLE001
            pass
        out = {
            "metric": "cg_member_matvecs_per_sec",
            "value": value,
            "unit": "member-matvecs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "step_ms": step_stats,
            "soak": soak,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "north_star headline (cfg3 operator, 1 rhs column): AddedDiag(LowRankRoot(C[512,8192,32]), "
                            "Diag(d[512,8192])) per GPU, linear_cg with rank-15 pivoted-Cholesky preconditioner, "
                            "cg_tolerance 1e-4 -> 11 iterations (floor)",
                "batch_per_gpu": B_PER_GPU, "N": N, "R": R, "rhs_columns": C_COLS, "precond_rank": RANK_K,
                "iterations": res.iterations, "matvecs_per_solve": matvecs_per_solve,
                "sharding": ((f"batch x{world}, no collective inside the solve; ONE all_gather of the solutions after the K "
                              "steps, inside the timed region (--gather end; the per-solve overlapped gather is timed "
                              "beside it: gather_modes)") if args.gather == "end" else
                             (f"batch x{world}, one all_gather of the solutions per solve, issued asynchronously and "
                              "overlapped with the next solve; resident kernels leave "
                              f"{os.environ.get('LO_OC_RESERVE_CUS', '0')} CUs' worth of slots to RCCL (--gather step)"))
                if world > 1 else "single GPU",
            },
            "value_kind": ("cg-iteration-equivalents (operator read once per solve): members x the 11 operator "
                           "applications the reference's linear_cg performs / wall time of the solve.  The timed engine "
                           "(k_cg_rspace) reads C from HBM ONCE per solve and carries the 11 iterations on R + 1 Krylov "
                           "coordinates in fp64: no product with C is executed per iteration.  The engine that applies "
                           "the operator 11 times on the rows is `operator_applications_per_sec`; ONE application of the "
                           "operator as an HBM pass (C read once) is `hbm_matvec`")
                          if engine_timed.startswith("resident R-space") else "operator applications on the rows of C",
            "operator_applications_per_sec": (matvec_engine or {}).get("value"),
            "operator_applications_engine": (matvec_engine or {}).get("engine"),
            "hbm_matvec": hbm_matvec,
            "roofline": {"bound": "hbm", "kernel": dom_kernel, "prof_scope": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_frac": (traffic / avg_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "achievable_peak": copy_gbs or HBM_ACHIEVABLE_GBS,
                         "frac_of_achievable": achieved / (copy_gbs or HBM_ACHIEVABLE_GBS),
                         "triad_this_box": triad_gbs, "copy_this_box": copy_gbs,
                         "achievable_peak_source": "copy_this_box when measured (1 GiB float4 copy of this run), else "
                                                   "the MI355X guide's 6.29 TB/s",
                         "algorithmic_bytes_per_launch": alg, "avg_launch_us": avg_s * 1e6, "launches_timed": cnt,
                         "algorithmic_bytes_survey": mv_alg,
                         "algorithmic_bytes_survey_note": "SURVEY 8(d): 4 (N R + N + 2 N c) per member and matvec = 587.2 MB "
                                                          "per batched matvec; the kernel needs these bytes ONCE per solve "
                                                          "plus its fp64 R x R inputs (algorithmic_bytes_per_launch)",
                         "note": ("per-launch compulsory bytes (C, 1/d, rhs in, x out, three fp64 R x R matrices and the "
                                  "eigenvalues of the diagonal form) / launch time.  The kernel holds a member's rows of C "
                                  "in VGPRs (16-byte chunk per lane, no transposition) between its one reduction over the "
                                  "rows and x = D^-1 (xi b + C y); the 11 iterations of the floor run as the CG of a DIAGONAL "
                                  "matrix on R + 1 coordinates in fp64 (one reduction of three values per iteration).  64 "
                                  "members are on chip at a time (235 VGPRs: C 128 + the fp64 chain) and a member takes "
                                  "~19 us from its first load to its last store (load 4 - 10, reduction + group all-reduce "
                                  "5 - 6, iterations 5.3, x 1.4 - 2: LO_OC_DEBUG stamps, profiles/r06), so 64 MB / 19 us = "
                                  "3.4 TB/s: bound by that latency chain at the on-chip capacity, not by HBM") if rs_diag else
                                 ("per-launch compulsory bytes (C, 1/d, rhs in, x out, the six fp64 R x R matrices of the "
                                  "R-space form) / launch time.  The kernel holds a member's rows of C in VGPRs between "
                                  "its one reduction over the rows and x = D^-1 (xi b + C y); the 11 iterations of the "
                                  "floor run on R + 1 coordinates in fp64.  64 members are on chip at a time and a "
                                  "member takes ~24 us from its first load to its last store (load 5, reduction + group "
                                  "all-reduce 5 - 9, iterations 9, x 1.5: LO_OC_DEBUG stamps, profiles/r05), so "
                                  "64 MB / 24 us = 2.7 TB/s: bound by that latency chain at the on-chip capacity, not by "
                                  "HBM") if rs_engine else
                                 ("per-launch compulsory bytes (C, d, 1/d, rhs in, x out, root-form preconditioner "
                                  "matrices; the kernel keeps the operator on chip for all 11 iterations) / launch time.  "
                                  "The kernel is bound by VALU issue and the latency of its per-iteration group "
                                  "all-reduce, not by HBM") if dom == "cg_onchip" else ""},
            "equivalent_streaming_rate": equiv,
            "matvec_equivalent": {"algorithmic_bytes": mv_alg, "avg_us": mv_s * 1e6,
                                  "equivalent_GBs": mv_alg / mv_s / 1e9, "north_star_budget_us": 122.0, "note": mv_note},
            "solves_per_sec_end_to_end": total_members / e2e,
            "end_to_end_ms": e2e * 1e3,
            "end_to_end_path": "A.solve(rhs) through the operator API, preconditioner memo cleared before every solve: "
                               "pivoted Cholesky + root-form preconditioner + CG in ONE resident launch "
                               "(lo_solve_fused_f32)",
            "end_to_end_solve_rel_err": woodbury_fp64_rel_err(Cm, d, rhs, x_e2e),
            "end_to_end_three_launch_ms": e2e3 * 1e3,
            "kernels": kernels,
            "final_mean_residual": res.mean_residual,
            "solve_rel_err": solve_rel_err,
            "timed_result_bitwise_equals_checked_result": timed_equals_checked,
            "first_solve_of_the_cache": {"solve_rel_err": solve_rel_err_first, "bitwise_equals_timed_result": first_equals_timed,
                                         "note": "the validation solve in front of the soak (a cache's first solves run the "
                                                 "dense R-space form, DESIGN 4.14); solve_rel_err above is the LAST TIMED "
                                                 "step's result against the fp64 closed form, and one more solve on the "
                                                 "timed engine reproduces it bit for bit"},
            "solve_rel_err_note": "max over the 512 members of ||x - x*|| / ||x*||, x* = fp64 Woodbury closed form of the "
                                  "same systems (north_star bar 1e-4); logdet rel-err with identical probes: "
                                  "cpu_baseline.parity_sample",
            "rccl_ranks": world if use_dist else 0,
            "per_rank_ms": per_rank_ms,
            "allgather_ms": allgather_ms,
            "gather_modes": gather_modes,
            "engine": engine_timed,
            "engine_per_rank": engines,
            "resident_timeouts": sum(g_["timeouts"] for g_ in gates),
            "resident_gate_per_rank": gates,
            "matvec_engine": matvec_engine,
            "rspace_dense_engine": rspace_dense_engine,
        }
        if cache_use and "error" not in cache_use and rspace_dense_engine:
            t_diag, t_dense = elapsed / args.steps * 1e3, rspace_dense_engine["ms_per_step"]
            t_eig, due = rspace_dense_engine["eigform_build_ms"], 25

            def per_solve(n, with_build):
                tot = (cache_use["cache_build_ms"] if with_build else (cache_use["rs_gram64_ms"] or 0.0))
                tot += t_dense * min(n, due - 1)
                if n >= due:
                    tot += t_eig + t_diag * (n - due + 1)
                return tot / n

            out["solve_ms_at_cache_use"] = {
                str(n): {"ms_per_solve": per_solve(n, False), "ms_per_solve_incl_cache_build": per_solve(n, True)}
                for n in (1, 25, 100)}
            out["solve_ms_at_cache_use"]["pieces_ms"] = {
                "solve_dense_form": t_dense, "solve_diagonal_form": t_diag, "rs_eigform_build": t_eig,
                "rs_gram64": cache_use["rs_gram64_ms"], "cache_build": cache_use["cache_build_ms"],
                "diagonal_form_from_solve": due}
            out["solve_ms_at_cache_use"]["note"] = (
                "wall ms per single-column solve when ONE preconditioner cache serves n solves.  ms_per_solve counts what the "
                "R-space engine adds to a cache (k_rs_gram64 once; k_rs_eigform once at the 25th solve, the dense form before "
                "it); ms_per_solve_incl_cache_build adds the whole cache build (pivoted Cholesky + root form + Gram matrices). "
                "`ms_per_step` of this line is the diagonal-form solve alone; a training loop that rebuilds the operator every "
                "step pays `end_to_end_ms`")
        elif cache_use:
            out["solve_ms_at_cache_use"] = cache_use
        if args.inject_timeouts > 0:
            out["inject_timeouts"] = {"injected": args.inject_timeouts, "soak_engines": engines_seen,
                                      "note": "engines of the soak's solves after the injected hand-off timeouts: "
                                              "cool-down on the streaming engine, then the resident kernel again"}
        # the one-launch end-to-end kernel at the headline shape: live HIP-event time, compulsory bytes per launch
        with lo_settings.cg_tolerance(TOL):
            prof_e = _profiled(e2e_api)
        clear_preconditioner_memo()
        fused_roof = None
        if "solve_fused" in prof_e:
            fused_roof = _roof("k_solve_fused<32,8,false>", "headline shape end to end: 512 members, rank-15 pivoted "
                               "Cholesky + root-form preconditioner + 11 CG iterations in ONE launch", prof_e["solve_fused"],
                               "hbm", 4 * B_PER_GPU * (N * (R + 3) + 3 * R * R + N),
                               "compulsory bytes per launch: C, d, rhs in; x, 1/d, F, EF, E out (L is never written).  "
                               "Bound by the latency chain of 15 pivot exchanges + 12 CG all-reduces per member at two "
                               "waves per SIMD (C in 128 VGPRs per lane), not by HBM",
                               _committed_traffic("traffic_fused.json"))
            out["end_to_end_kernel"] = fused_roof
        if extras is not None:
            out["other_configs"], out["rooflines"] = extras
            if fused_roof is not None:
                out["rooflines"].insert(0, fused_roof)
            if hbm_matvec and "error" not in hbm_matvec:
                out["rooflines"].insert(0, hbm_matvec)
            out["order"] = ("other_configs (cfg2 - cfg5 extras) -> HBM ceilings -> validation solve -> W warm-up steps -> "
                            "K timed steps -> soak -> end-to-end / roofline measurements -> cpu_baseline")
        elif extras_error is not None:
            out["other_configs_error"] = extras_error
        if strong:
            out["other_configs"] = strong
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(device=device)
            if not args.no_extras:
                torch.cuda.empty_cache()
                try:  # (the headline line must not be lost to an extra)
                    out["config_baselines"] = config_baselines(device)
                except Exception as e:  # noqa: BLE001
                    out["config_baselines_error"] = repr(e)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.close(real_stdout)


if __name__ == "__main__":
    main()
