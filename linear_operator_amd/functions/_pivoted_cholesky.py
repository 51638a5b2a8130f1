"""PivotedCholesky Function (reference: linear_operator/functions/_pivoted_cholesky.py:12-147), forward on the
device through csrc/lo_pivchol.hip (rows generated from the operator descriptor instead of the generic
__getitem__ / gather / scatter chain).  Operators without a descriptor (opaque trees) are evaluated to a
dense tensor first."""
from __future__ import annotations

import torch
from torch.autograd import Function

from .. import kernels as K
from .. import settings
from ._common import not_yet


class PivotedCholesky(Function):
    @staticmethod
    def forward(ctx, representation_tree, max_iter, error_tol, *matrix_args):
        matrix = representation_tree(*matrix_args)
        if error_tol is None:
            error_tol = settings.preconditioner_tolerance.value()
        if settings.verbose_linalg.on():
            settings.verbose_linalg.logger.debug(
                f"Running Pivoted Cholesky on a {matrix.shape} RHS for {max_iter} iterations."
            )
        desc = matrix._kernel_descriptor()
        if desc is not None and desc.diag_mode != 0:
            desc = None  # the kernel factors descriptor WITHOUT its diagonal; a genuine A + D goes dense
        if desc is None:
            dense = matrix.to_dense()
            desc = K.dense_diag_descriptor(dense, None)
        L, perm = K.pivoted_cholesky(desc, max_iter, float(error_tol))
        ctx.mark_non_differentiable(perm)
        return L, perm

    @staticmethod
    def backward(ctx, grad_output, _):
        not_yet("PivotedCholesky")
