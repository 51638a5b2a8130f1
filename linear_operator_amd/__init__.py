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


def install_as(name: str = "linear_operator", force: bool = False):
    """Make this package importable under another name -- `install_as("linear_operator")` turns it into the drop-in
    the reference's users (GPyTorch: `import linear_operator`, `from linear_operator.operators import ...`,
    `linear_operator.settings.cg_tolerance(...)`, reference linear_operator/__init__.py:4-18) import, with the very same
    module OBJECTS: rebinding `linear_operator.utils.linear_cg` -- the solver seam the reference's tests patch,
    linear_operator/test/linear_operator_test_case.py:555-556 -- rebinds `linear_operator_amd.utils.linear_cg`.

    Every submodule is imported and aliased in `sys.modules` (`linear_operator.operators.added_diag_linear_operator`,
    `linear_operator.utils.lanczos`, ...).  Refuses to shadow a different package already imported under `name`
    unless `force` is set.  Returns the package."""
    import importlib
    import pkgutil
    import sys

    me = sys.modules[__name__]
    other = sys.modules.get(name)
    if other is not None and other is not me and not force:
        raise ImportError(f"a different package is already imported as {name!r}: {getattr(other, '__file__', other)}; "
                          "pass force=True to shadow it")
    for info in pkgutil.walk_packages(me.__path__, prefix=__name__ + "."):
        importlib.import_module(info.name)
    for full, mod in list(sys.modules.items()):
        if mod is not None and (full == __name__ or full.startswith(__name__ + ".")):
            sys.modules[name + full[len(__name__):]] = mod
    return me

__all__ = [
    "LinearOperator", "to_dense", "to_linear_operator", "operators", "settings", "utils",
    "add_diagonal", "add_jitter", "diagonal", "diagonalization", "inv_quad", "inv_quad_logdet", "logdet", "matmul", "pivoted_cholesky",
    "solve", "sqrt_inv_matmul", "install_as",
]
