"""InvQuadLogdet Function (reference: linear_operator/functions/_inv_quad_logdet.py:13-161), forward:
probes from the preconditioner distribution, ONE preconditioned CG call for [probes | rhs] with the CG
tridiagonals of the probe columns, SLQ logdet, inv_quad = sum_rows solves o rhs.  The tridiagonal
eigendecomposition + quadrature is one kernel on the device (csrc/lo_eig.hip) instead of the reference's
device -> host -> device round trip (utils/lanczos.py:179-189)."""
from __future__ import annotations

import torch
from torch.autograd import Function

from .. import kernels as K
from .. import settings
from ..utils.lanczos import lanczos_tridiag_to_diag
from ..utils.stochastic_lq import StochasticLQ
from ._common import not_yet


class InvQuadLogdet(Function):
    @staticmethod
    def forward(ctx, representation_tree, precond_representation_tree, preconditioner, num_precond_args, inv_quad,
                probe_vectors, probe_vector_norms, *args):
        inv_quad_rhs = None
        if inv_quad:
            inv_quad_rhs, args = args[0], args[1:]
        if num_precond_args:
            matrix_args, precond_args = args[:-num_precond_args], args[-num_precond_args:]
        else:
            matrix_args, precond_args = args, tuple()
        linear_op = representation_tree(*matrix_args)
        precond_lt = precond_representation_tree(*precond_args)
        dtype, device = linear_op.dtype, linear_op.device
        matrix_shape, batch_shape = linear_op.matrix_shape, linear_op.batch_shape

        if probe_vectors is None or probe_vector_norms is None:  # reference :78-110
            if settings.deterministic_probes.on():
                raise NotImplementedError("deterministic_probes is deprecated in the reference and not implemented")
            num_random_probes = settings.num_trace_samples.value()
            probe_vectors = precond_lt.zero_mean_mvn_samples(num_random_probes)  # [P, *batch, N]
            probe_vectors = probe_vectors.unsqueeze(-2).transpose(0, -2).squeeze(0).mT.contiguous()  # [*batch, N, P]
            probe_vector_norms = torch.linalg.vector_norm(probe_vectors, ord=2, dim=-2, keepdim=True)
            probe_vectors = probe_vectors.div(probe_vector_norms)

        rhs_list = [probe_vectors]
        num_random_probes = probe_vectors.size(-1)
        num_inv_quad_solves = 0
        if inv_quad:
            if inv_quad_rhs.ndimension() == 1:
                inv_quad_rhs = inv_quad_rhs.unsqueeze(-1)
            rhs_list.append(inv_quad_rhs)
            num_inv_quad_solves = inv_quad_rhs.size(-1)
        rhs = torch.cat(rhs_list, -1)
        solves, t_mat = linear_op._solve(rhs, preconditioner, num_tridiag=num_random_probes)  # reference :133

        logdet_term = torch.zeros(batch_shape, dtype=dtype, device=device)
        inv_quad_term = torch.zeros(batch_shape, dtype=dtype, device=device)
        if settings.skip_logdet_forward.off():
            if torch.any(torch.isnan(t_mat)).item():  # reference :141-142
                logdet_term = torch.tensor(float("nan"), dtype=dtype, device=device)
            elif t_mat.is_cuda and t_mat.dtype == torch.float32 and t_mat.size(-1) <= 32:
                _, _, logdet_term = K.tridiag_eigh_slq(t_mat, matrix_shape[-1])
                logdet_term = logdet_term.reshape(batch_shape)
            else:
                evals, evecs = lanczos_tridiag_to_diag(t_mat)
                (logdet_term,) = StochasticLQ().to_dense(matrix_shape, evals, evecs, [lambda x: x.log()])
        if inv_quad:  # reference :151-153
            inv_quad_solves = solves.narrow(-1, num_random_probes, num_inv_quad_solves)
            inv_quad_term = (inv_quad_solves * inv_quad_rhs).sum(-2)
        return inv_quad_term, logdet_term

    @staticmethod
    def backward(ctx, inv_quad_grad_output, logdet_grad_output):
        not_yet("InvQuadLogdet")
