"""csrc/lo_probes.hip: the host glue of InvQuadLogdet as kernels.
  * `kernels.probe_vectors` against the expressions of the reference's forward (functions/_inv_quad_logdet.py:91-110,
    :131: zero_mean_mvn_samples of the preconditioner L L^T + D, column norms, division, cat) evaluated with torch in
    float64 on the same draws;
  * `kernels.iql_backward_factors` against the element-wise part of the reference's backward (:183-213);
  * through the operator API: forward + backward of `inv_quad_logdet` with the kernels (drawn probes) equals the same
    call with the SAME probes injected (which takes the torch path of the forward), gradients included.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from linear_operator_amd import kernels as K  # noqa: E402
from linear_operator_amd import settings  # noqa: E402
from linear_operator_amd.functions import _inv_quad_logdet as iql_mod  # noqa: E402
from linear_operator_amd.operators import AddedDiagLinearOperator, DiagLinearOperator, LowRankRootLinearOperator  # noqa: E402


@pytest.mark.parametrize("B,N,k,P,q,layout,const", [
    (3, 1000, 15, 16, 1, "rows", False),   # L as the [B, k, N] rows the factorisation writes (a strided view)
    (2, 777, 7, 10, 0, "cols", False),     # GPyTorch's default probe count, no inv_quad columns
    (1, 300, 3, 5, 3, "cols", True),       # constant diagonal, several inv_quad columns
    (4, 4096, 15, 16, 1, "bcast", False),  # one L shared by the batch
])
def test_probe_vectors_match_the_reference_expressions(B, N, k, P, q, layout, const):
    g = torch.Generator(device="cuda").manual_seed(5 + N)
    if layout == "rows":
        L = torch.randn(B, k, N, generator=g, device="cuda").mT
    elif layout == "bcast":
        L = torch.randn(1, N, k, generator=g, device="cuda").expand(B, N, k)
    else:
        L = torch.randn(B, N, k, generator=g, device="cuda")
    d = (torch.rand(B, 1, generator=g, device="cuda") + 0.5) if const else (torch.rand(B, N, generator=g, device="cuda") + 0.5)
    e1 = torch.randn(B, k, P, generator=g, device="cuda")
    e2 = torch.randn(B, N, P, generator=g, device="cuda")
    iq = torch.randn(B, N, q, generator=g, device="cuda") if q else None
    rhs, norms = K.probe_vectors(L, d, e1, e2, iq, (B,))
    z = L.double() @ e1.double() + d.double().expand(B, N).sqrt().unsqueeze(-1) * e2.double()
    nr = z.norm(dim=-2, keepdim=True)
    assert rhs.shape == (B, N, P + q) and norms.shape == (B, 1, P)
    assert torch.allclose(norms.double(), nr, rtol=1e-5)
    assert torch.allclose(rhs[..., :P].double(), z / nr, rtol=1e-4, atol=1e-6)
    if q:
        assert torch.equal(rhs[..., P:], iq)


@pytest.mark.parametrize("B,N,P,q", [(3, 1000, 16, 1), (2, 333, 10, 0), (1, 5000, 4, 3)])
def test_backward_factors_match_the_reference_expressions(B, N, P, q):
    g = torch.Generator(device="cuda").manual_seed(9 + N)
    solves = torch.randn(B, N, P + q, generator=g, device="cuda")
    pp = torch.randn(B, N, P + q, generator=g, device="cuda")  # (the preconditioner applied to the whole block)
    norms = torch.rand(B, 1, P, generator=g, device="cuda") + 0.5
    g_ld = torch.randn(B, generator=g, device="cuda")
    g_iq = torch.randn(B, q, generator=g, device="cuda") if q else None
    left, right, pl, pr = K.iql_backward_factors(solves, pp, norms, g_ld, g_iq, P)
    coef = 1.0 / P
    ppv = pp[..., :P] * norms  # P^-1 of the raw probes
    assert torch.allclose(left[..., :P], solves[..., :P] * norms * g_ld.view(B, 1, 1) * coef, rtol=1e-5, atol=1e-7)
    assert torch.allclose(right[..., :P], ppv, rtol=1e-6)
    assert torch.allclose(pl, -ppv * coef, rtol=1e-5, atol=1e-7) and torch.allclose(pr, ppv * g_ld.view(B, 1, 1), rtol=1e-5, atol=1e-7)
    if q:
        assert torch.allclose(left[..., P:], -solves[..., P:] * g_iq.unsqueeze(-2), rtol=1e-6)
        assert torch.equal(right[..., P:], solves[..., P:])


class _Probed(AddedDiagLinearOperator):
    _probes = None

    def _probe_vectors_and_norms(self):  # hook: reference _linear_operator.py:629-633
        return self._probes if self._probes is not None else (None, None)


def test_drawn_probes_through_the_kernels_equal_the_same_probes_injected(monkeypatch):
    B, N, R = 4, 3000, 16
    g = torch.Generator(device="cuda").manual_seed(31)
    C0 = torch.randn(B, N, R, generator=g, device="cuda") / R ** 0.5
    d0 = torch.rand(B, N, generator=g, device="cuda") + 0.5
    rhs0 = torch.randn(B, N, 1, generator=g, device="cuda")
    seen = {}
    orig = iql_mod._fused_probe_block

    def spy(*a, **kw):
        out = orig(*a, **kw)
        seen["block"] = out
        return out

    monkeypatch.setattr(iql_mod, "_fused_probe_block", spy)

    def run(inject):
        C = C0.clone().requires_grad_(True)
        d = d0.clone().requires_grad_(True)
        rhs = rhs0.clone().requires_grad_(True)
        A = _Probed(LowRankRootLinearOperator(C), DiagLinearOperator(d))
        if inject is not None:
            A._probes = inject
        with settings.cg_tolerance(1e-4), settings.num_trace_samples(16), settings.min_preconditioning_size(100):
            iq, ld = A.inv_quad_logdet(rhs, logdet=True)
        (iq.sum() + ld.sum()).backward()
        return iq.detach(), ld.detach(), C.grad, d.grad, rhs.grad

    a = run(None)
    assert seen.get("block") is not None, "the kernel path was not taken"
    block, norms = seen["block"]
    probes = block[..., :16].contiguous()
    b = run((probes, norms))
    for x, y in zip(a, b):
        assert torch.allclose(x, y, rtol=2e-4, atol=2e-5 * float(y.abs().max())), float((x - y).abs().max())
