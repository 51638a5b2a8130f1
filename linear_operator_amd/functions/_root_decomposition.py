"""RootDecomposition Function (reference: linear_operator/functions/_root_decomposition.py:11-102), forward on the
device: Lanczos (csrc/lo_lanczos.hip), eigendecomposition of the jittered tridiagonals (csrc/lo_eig.hip for k <= 32),
then one pass producing Q V o sqrt(lambda) and / or Q V / sqrt(lambda) (lo_root_from_lanczos_f32).  SURVEY 8(f) rank 2."""
from __future__ import annotations

import torch
from torch.autograd import Function

from .. import kernels as K
from .. import settings
from ..utils import lanczos


class RootDecomposition(Function):
    @staticmethod
    def forward(ctx, representation_tree, max_iter, dtype, device, batch_shape, matrix_shape, root, inverse,
                initial_vectors, *matrix_args):
        ctx.representation_tree = representation_tree
        ctx.inverse = bool(inverse)
        linear_op = representation_tree(*matrix_args)
        q_mat, t_mat = lanczos.lanczos_tridiag(  # :49-57
            linear_op._matmul, max_iter, dtype=dtype, device=device, matrix_shape=matrix_shape,
            batch_shape=batch_shape, init_vecs=initial_vectors,
        )
        if batch_shape is None:  # :59-61
            q_mat = q_mat.unsqueeze(-3)
            t_mat = t_mat.unsqueeze(-3)
        if t_mat.ndimension() == 3:  # one probe vector  :62-64
            q_mat = q_mat.unsqueeze(0)
            t_mat = t_mat.unsqueeze(0)
        n_probes = t_mat.size(0)

        mins = t_mat.diagonal(dim1=-2, dim2=-1).min(dim=-1, keepdim=True)[0].unsqueeze(-1)  # :67
        jitter_mat = (settings.tridiagonal_jitter.value() * mins) * torch.eye(
            t_mat.size(-1), device=t_mat.device, dtype=t_mat.dtype
        ).expand_as(t_mat)
        eigenvalues, eigenvectors = lanczos.lanczos_tridiag_to_diag(t_mat + jitter_mat)  # :71

        # q_mat <- q_mat V ; root = q_mat o sqrt(lambda) ; inverse = q_mat / sqrt(lambda)   :73-85
        q_mat, root_t, inv_t = K.root_from_lanczos(q_mat, eigenvectors, eigenvalues, want_root=bool(root),
                                                    want_inverse=bool(inverse))
        root_evals = eigenvalues.sqrt()
        empty = torch.empty(0, dtype=q_mat.dtype, device=q_mat.device)
        root_t = empty if root_t is None else root_t
        inv_t = empty if inv_t is None else inv_t

        if batch_shape is None:  # :90-94
            root_t = root_t.squeeze(1) if root_t.numel() else root_t
            q_mat = q_mat.squeeze(1)
            root_evals = root_evals.squeeze(1)
            inv_t = inv_t.squeeze(1) if inv_t.numel() else inv_t
        if n_probes == 1:  # :95-99
            root_t = root_t.squeeze(0) if root_t.numel() else root_t
            q_mat = q_mat.squeeze(0)
            root_evals = root_evals.squeeze(0)
            inv_t = inv_t.squeeze(0) if inv_t.numel() else inv_t
        ctx.save_for_backward(*matrix_args, q_mat, root_evals, inv_t)  # :100-101
        return root_t, inv_t

    @staticmethod
    def backward(ctx, root_grad_output, inverse_grad_output):  # reference :104-171
        """With R = Q sqrt(Lambda) and R_inv = Q / sqrt(Lambda):  dA = sym-contraction of
        (grad_R - R_inv grad_Rinv^T R_inv) with R_inv / 2, handed to `_bilinear_derivative` (csrc/lo_bilinear.hip)."""
        if not any(ctx.needs_input_grad):
            return tuple([None] * (9 + len(ctx.saved_tensors) - 3))

        def is_empty(t):
            return t is None or t.numel() == 0 or (t.numel() == 1 and t.reshape(-1)[0] == 0)

        root_grad_output = None if is_empty(root_grad_output) else root_grad_output
        inverse_grad_output = None if is_empty(inverse_grad_output) else inverse_grad_output
        *matrix_args, q_mat, root_evals, inverse = ctx.saved_tensors
        is_batch = False
        grads = []
        for g in (root_grad_output, inverse_grad_output):  # :125-138
            if g is not None:
                if (g.ndimension() == 2 and q_mat.ndimension() > 2) or (g.ndimension() == 3 and q_mat.ndimension() > 3):
                    g = g.unsqueeze(0)
                    is_batch = True
            grads.append(g)
        root_grad_output, inverse_grad_output = grads
        linear_op = ctx.representation_tree(*matrix_args)
        if not ctx.inverse:
            inverse = q_mat / root_evals.unsqueeze(-2)  # :147-148
        left_factor = torch.zeros_like(inverse)
        if root_grad_output is not None:
            left_factor = left_factor + root_grad_output
        if inverse_grad_output is not None:  # -R_inv grad^T R_inv  (:153-155)
            left_factor = left_factor - torch.matmul(inverse, inverse_grad_output.mT).matmul(inverse)
        right_factor = inverse.div(2.0)
        if is_batch:  # several probe vectors: the probe dimension joins the columns (:160-164)
            left_factor = left_factor.permute(1, 0, 2, 3).contiguous()
            left_factor = left_factor.view(inverse.size(1), -1, left_factor.size(-1))
            right_factor = right_factor.permute(1, 0, 2, 3).contiguous()
            right_factor = right_factor.view(inverse.size(1), -1, right_factor.size(-1))
        res = linear_op._bilinear_derivative(left_factor.contiguous(), right_factor.contiguous())
        return tuple([None] * 9 + list(res))


__all__ = ["RootDecomposition"]
