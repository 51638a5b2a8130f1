"""Autograd wrappers.  Matmul, Solve, InvQuad and InvQuadLogdet have forward and backward (more solves plus the
`_bilinear_derivative` contractions of csrc/lo_bilinear.hip), PivotedCholesky has the reference's
backward for dense and dense-root operators; RootDecomposition is forward-only: asking it for gradients fails loudly."""
from __future__ import annotations


def not_yet(name):
    raise NotImplementedError(
        f"{name}.backward is not implemented in linear_operator_amd yet (forward-only scope, SURVEY.md section 8(f) "
        "rank 1); detach the operator / right-hand side or wrap the call in torch.no_grad()."
    )
