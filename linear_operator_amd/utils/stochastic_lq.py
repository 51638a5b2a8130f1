"""StochasticLQ (reference: linear_operator/utils/stochastic_lq.py:8-82).  `to_dense` keeps the generic
list-of-functions interface (a handful of [batch, k] elementwise ops); the fused eigh + log quadrature the
logdet path uses lives in csrc/lo_eig.hip and is called directly by functions/_inv_quad_logdet.py."""
from __future__ import annotations

import torch

from .lanczos import lanczos_tridiag


class StochasticLQ:
    def __init__(self, max_iter=15, num_random_probes=10):
        self.max_iter = max_iter
        self.num_random_probes = num_random_probes

    def lanczos_batch(self, matmul_closure, rhs_vectors):
        return lanczos_tridiag(
            matmul_closure, self.max_iter, init_vecs=rhs_vectors, dtype=rhs_vectors.dtype,
            device=rhs_vectors.device, batch_shape=rhs_vectors.shape[-2:],
            matrix_shape=torch.Size((rhs_vectors.size(-2), rhs_vectors.size(-2))),
        )

    def to_dense(self, matrix_shape, eigenvalues, eigenvectors, funcs):
        """tr(f(A)) ~= (n / P) sum_p sum_i (e_1^T v_{p,i})^2 f(lambda_{p,i}) for each f in funcs."""
        batch_shape = torch.Size(eigenvalues.shape[1:-1])
        n_probes = eigenvalues.size(0)
        weights = eigenvectors[..., 0, :].pow(2)  # first components, [P, *batch, k]
        scale = matrix_shape[-1] / float(n_probes)
        results = []
        for func in funcs:
            acc = torch.zeros(batch_shape, dtype=eigenvalues.dtype, device=eigenvalues.device)
            for j in range(n_probes):
                acc = acc + scale * (weights[j] * func(eigenvalues[j])).sum(-1)
            results.append(acc)
        return results
