"""linear_operator_amd -- MI355X-native implementation of the iterative solve / logdet hot path of
cornellius-gp/linear_operator (batched preconditioned CG, SLQ logdet, pivoted-Cholesky preconditioner,
Lanczos, and the structured matvecs that feed them), behind the reference's own operator API.

The compute backend is liblo_amd.so (hand-written gfx950 kernels, C ABI in include/lo_amd.h); there is no
CPU fallback for the iterative path.  See DESIGN.md / INTEGRATION.md.
"""
from . import operators, settings, utils
from .functions import (add_diagonal, add_jitter, diagonal, diagonalization, inv_quad, inv_quad_logdet, logdet, matmul,
                        pivoted_cholesky, solve, sqrt_inv_matmul)
from .operators import LinearOperator, to_dense, to_linear_operator

__version__ = "0.1.0"

__all__ = [
    "LinearOperator", "to_dense", "to_linear_operator", "operators", "settings", "utils",
    "add_diagonal", "add_jitter", "diagonal", "diagonalization", "inv_quad", "inv_quad_logdet", "logdet", "matmul", "pivoted_cholesky",
    "solve", "sqrt_inv_matmul",
]
