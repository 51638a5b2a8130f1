"""Diagonalization Function (reference: linear_operator/functions/_diagonalization.py:11-60), forward: device Lanczos
(csrc/lo_lanczos.hip), jitter, eigendecomposition of the k x k matrices, Q <- Q V.  SURVEY 8(f) rank 2.
The reference's jitter term is `diag_embed(jitter * mins).expand_as(t_mat)` with `mins` of trailing size 1: a
[..., 1, 1] tensor broadcast over the WHOLE k x k matrix (:48-50), so the jittered matrix is no longer tridiagonal;
it is reproduced as written and handed to a dense symmetric eigensolver (ATen, k <= max_root_decomposition_size)."""
from __future__ import annotations

import torch
from torch.autograd import Function

from .. import settings
from ..utils import lanczos


class Diagonalization(Function):
    @staticmethod
    def forward(ctx, representation_tree, device, dtype, matrix_shape, max_iter, batch_shape, *matrix_args):
        linear_op = representation_tree(*matrix_args)
        q_mat, t_mat = lanczos.lanczos_tridiag(  # :33-40
            linear_op._matmul, max_iter, dtype=dtype, device=device, matrix_shape=matrix_shape,
            batch_shape=batch_shape,
        )
        if batch_shape is None:  # :42-44
            q_mat = q_mat.unsqueeze(-3)
            t_mat = t_mat.unsqueeze(-3)
        if t_mat.ndimension() == 3:  # one probe vector  :45-47
            q_mat = q_mat.unsqueeze(0)
            t_mat = t_mat.unsqueeze(0)
        mins = torch.diagonal(t_mat, dim1=-1, dim2=-2).min(dim=-1, keepdim=True)[0]  # :49
        jitter_mat = torch.diag_embed(settings.tridiagonal_jitter.value() * mins).expand_as(t_mat)  # :50-51
        eigenvalues, eigenvectors = lanczos.lanczos_tridiag_to_diag(t_mat + jitter_mat, tridiagonal=False)  # :52
        q_mat = q_mat.matmul(eigenvectors)  # :55
        if batch_shape is None:
            q_mat = q_mat.squeeze(1)
        q_mat = q_mat.squeeze(0)
        eigenvalues = eigenvalues.squeeze(0)
        ctx.save_for_backward(*matrix_args, q_mat, eigenvalues)
        return eigenvalues, q_mat

    @staticmethod
    def backward(ctx, evals_grad_output, evecs_grad_output):  # reference :62-88
        """Explicit eigendecomposition gradients (Ionescu et al. 2015) as a DENSE dL/dM -- like the reference, which
        returns it in the slot of one dense matrix argument (:86)."""
        q_mat, eigenvalues = ctx.saved_tensors[-2], ctx.saved_tensors[-1]
        # (K~)_ij = 1_{i != j} / (sigma_i - sigma_j), a little jitter against zeros
        kmat = (eigenvalues.unsqueeze(-1) - eigenvalues.unsqueeze(-2) + 1e-10).reciprocal()
        torch.diagonal(kmat, dim1=-1, dim2=-2).zero_()
        inner_term = kmat.mT * q_mat.mT.matmul(evecs_grad_output)  # dU = U (K~^T o (U^T dL/dU)) U^T
        term1 = q_mat.matmul(inner_term).matmul(q_mat.mT)
        term2 = q_mat.matmul(torch.diag_embed(evals_grad_output)).matmul(q_mat.mT)  # dSigma = U dL/dSigma U^T
        return tuple([None] * 6 + [term1 + term2])


__all__ = ["Diagonalization"]
