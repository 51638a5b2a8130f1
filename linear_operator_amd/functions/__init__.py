"""Functional API (reference: linear_operator/functions/__init__.py:17-271): thin wrappers -> methods."""
from __future__ import annotations

import torch

from ..operators import LinearOperator, to_linear_operator


def _op(x):
    return x if isinstance(x, LinearOperator) else to_linear_operator(x)


def add_diagonal(input, diag):
    return _op(input).add_diagonal(diag)


def add_jitter(input, jitter_val: float = 1e-3):
    return _op(input).add_jitter(jitter_val=jitter_val)


def diagonal(input):
    return _op(input).diagonal()


def diagonalization(input, method=None):
    return _op(input).diagonalization(method=method)


def inv_quad(input, inv_quad_rhs, reduce_inv_quad: bool = True):
    return _op(input).inv_quad(inv_quad_rhs, reduce_inv_quad=reduce_inv_quad)


def inv_quad_logdet(input, inv_quad_rhs=None, logdet: bool = False, reduce_inv_quad: bool = True):
    return _op(input).inv_quad_logdet(inv_quad_rhs=inv_quad_rhs, logdet=logdet, reduce_inv_quad=reduce_inv_quad)


def logdet(input):
    return _op(input).logdet()


def matmul(input, other):
    return _op(input).matmul(other)


def pivoted_cholesky(input, rank: int, error_tol=None, return_pivots: bool = False):
    return _op(input).pivoted_cholesky(rank=rank, error_tol=error_tol, return_pivots=return_pivots)


def sqrt_inv_matmul(input, rhs, lhs=None):
    return _op(input).sqrt_inv_matmul(rhs, lhs)


def solve(input, rhs, lhs=None):
    return _op(input).solve(right_tensor=rhs, left_tensor=lhs)


__all__ = ["add_diagonal", "add_jitter", "diagonal", "diagonalization", "inv_quad", "inv_quad_logdet", "logdet", "matmul",
           "pivoted_cholesky", "solve", "sqrt_inv_matmul"]
