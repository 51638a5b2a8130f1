"""Helpers of the iterative solve / logdet path.  `linear_cg` is looked up through this module attribute at
call time by LinearOperator._solve, so it can be wrapped / replaced exactly as in the reference
(linear_operator/utils/__init__.py; seam used by linear_operator/test/linear_operator_test_case.py:555-556)."""
from . import broadcasting, cholesky, errors, lanczos, permutation, stochastic_lq, warnings
from .contour_integral_quad import contour_integral_quad
from .linear_cg import linear_cg
from .minres import minres
from .stochastic_lq import StochasticLQ

__all__ = ["contour_integral_quad", "broadcasting", "cholesky", "errors", "lanczos", "linear_cg", "minres", "permutation", "stochastic_lq",
           "StochasticLQ", "warnings"]
