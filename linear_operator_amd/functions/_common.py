"""Forward-only autograd wrappers: the reference's Functions save tensors for a backward pass built from more
solves and `_bilinear_derivative` (SURVEY 8(b) 'Autograd', 8(f) rank 1 -- 'next').  The wrappers are kept so
that backward can be added behind the same call sites; until then asking for gradients fails loudly."""
from __future__ import annotations


def not_yet(name):
    raise NotImplementedError(
        f"{name}.backward is not implemented in linear_operator_amd yet (forward-only scope, SURVEY.md section 8(f) "
        "rank 1); detach the operator / right-hand side or wrap the call in torch.no_grad()."
    )
