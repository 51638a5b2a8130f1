"""The gate of the operator-resident kernels (lo_resident_status, csrc/lo_cg.hip): an injected hand-off timeout is redone on
the streaming engine, starts a cool-down of 16 entry-point calls, and the resident kernel is back afterwards."""
import numpy as np
import pytest
import torch

import cases
from conftest import load_golden, max_rel_err_cols

pytestmark = pytest.mark.gpu

from linear_operator_amd import kernels as K  # noqa: E402
from oracle import lo_oracle as orc  # noqa: E402  (the checker)
from oracle import lo_oracle_c as occ  # noqa: E402  (the checker, C restatement)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def host(t):
    return t.detach().cpu().numpy()


def _precond(desc, d_t):
    L, perm = K.pivoted_cholesky(desc, 15)
    return K.precond_build(L, d_t, constant_diag=False, root=desc.A0, perm=perm), perm


# --------------------------------------------------------------------------------------- gate of the resident kernels
def test_resident_gate_cools_down_and_rearms():
    C, d, rhs = cases.lowrank_diag(5501, 70, 4096, 32, 1)
    desc = K.lowrank_diag_descriptor(dev(C), dev(d))
    pre, _ = _precond(desc, dev(d))
    K.set_onchip_cg(True)  # (ends any cool-down another test may have left)
    ref = K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-4)
    assert K.cg_last_executed()["resident"], "the resident kernel must take this shape"
    s0 = K.resident_status()
    assert s0["cooldown"] == 0 and not s0["user_disabled"]
    try:
        K.inject_resident_timeouts(1)
        hit = K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-4)  # timed out inside, redone by the streaming engine
        e = K.cg_last_executed()
        s1 = K.resident_status()
        assert not e["resident"] and e["streaming_iterations"] >= 11
        assert s1["timeouts"] == s0["timeouts"] + 1 and s1["cooldown"] == s0["backoff"] == 16 and s1["backoff"] == 32
        assert hit.iterations == ref.iterations and max_rel_err_cols(host(hit.x), host(ref.x)) < 2e-5
        engines = []
        for _ in range(s1["cooldown"]):
            r = K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-4)
            engines.append(bool(K.cg_last_executed()["resident"]))
            assert max_rel_err_cols(host(r.x), host(ref.x)) < 2e-5
        # calls 1 .. 15 of the cool-down on the streaming engine, the 16th re-arms and runs resident again
        assert engines == [False] * 15 + [True], engines
        s2 = K.resident_status()
        assert s2["cooldown"] == 0 and s2["rearms"] == s1["rearms"] + 1 and s2["timeouts"] == s1["timeouts"]
        assert s2["backoff"] == 16, "a clean resident solve makes the next cool-down short again"
        # the pivoted Cholesky's resident kernel obeys the same gate
        K.inject_resident_timeouts(1)
        K.cg_solve(desc, dev(rhs), precond=pre, tolerance=1e-4)
        K._hip.prof_enable(True)
        L1, p1 = K.pivoted_cholesky(desc, 15)
        torch.cuda.synchronize()
        prof = K._hip.prof_report()
        K._hip.prof_enable(False)
        assert "pc_update" in prof, sorted(prof)
        K.set_onchip_cg(True)  # ends the cool-down at once
        assert K.resident_status()["cooldown"] == 0
        L0, p0 = K.pivoted_cholesky(desc, 15)
        assert torch.equal(L0, L1) and torch.equal(p0, p1)
    finally:
        K.inject_resident_timeouts(0)
        K.set_onchip_cg(True)


def test_resident_matvec_obeys_the_gate():
    """lo_matvec_f32 is asynchronous: a launch of k_lr_mv that loses its co-residency repairs itself INSIDE the kernel (the
    workgroups recompute C^T v of their member from HBM) and leaves its tag in a pinned error word; the NEXT call sees it,
    starts the cool-down of the resident kernels and runs the two streaming passes until the gate re-arms."""
    C, d, v = cases.lowrank_diag(5601, 40, 8192, 32, 1)
    desc = K.lowrank_diag_descriptor(dev(C), dev(d))
    vd = dev(v)

    def mv():
        torch.cuda.synchronize()
        K._hip.prof_enable(True)
        try:
            y = K.matvec(desc, vd)
            torch.cuda.synchronize()
            return y, K._hip.prof_report()
        finally:
            K._hip.prof_enable(False)

    K.set_onchip_cg(True)
    y0, p0 = mv()
    assert "lr_mv" in p0
    s0 = K.resident_status()
    try:
        K.inject_resident_timeouts(1)
        y1, p1 = mv()                       # every workgroup "lost": repaired in the kernel, result to summation order
        assert "lr_mv" in p1 and max_rel_err_cols(host(y1), host(y0)) < 2e-6
        assert K.resident_status()["timeouts"] == s0["timeouts"]  # nobody has looked at the error word yet
        y2, p2 = mv()                       # ... this call does: cool-down, two streaming passes
        s2 = K.resident_status()
        assert "lr_mv" not in p2 and s2["timeouts"] == s0["timeouts"] + 1 and s2["cooldown"] > 0
        assert max_rel_err_cols(host(y2), host(y0)) < 2e-6
        seen = []
        for _ in range(s2["cooldown"] + 1):
            _, p = mv()
            seen.append("lr_mv" in p)
        assert seen[-1] and not any(seen[:-2]), seen  # back on the resident kernel once the cool-down is served
        assert torch.equal(K.matvec(desc, vd), y0)
    finally:
        K.inject_resident_timeouts(0)
        K.set_onchip_cg(True)
