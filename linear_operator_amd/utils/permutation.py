"""apply_permutation / inverse_permutation (reference: linear_operator/utils/permutation.py:9-102).
The pivoted-Cholesky kernel generates permuted rows on the fly from the operator descriptor; these helpers
remain for API parity (tests, callers that post-process pivots) and work on dense tensors / operators."""
from __future__ import annotations

import torch


def apply_permutation(matrix, left_permutation=None, right_permutation=None):
    """Pi_left K Pi_right^T for (batched, possibly partial) permutation vectors."""
    dense = matrix.to_dense() if hasattr(matrix, "to_dense") and not torch.is_tensor(matrix) else matrix
    if left_permutation is None and right_permutation is None:
        return dense
    batch_shape = dense.shape[:-2]
    if left_permutation is None:
        left_permutation = torch.arange(dense.size(-2), device=dense.device)
    if right_permutation is None:
        right_permutation = torch.arange(dense.size(-1), device=dense.device)
    lp = left_permutation.expand(*batch_shape, left_permutation.size(-1))
    rp = right_permutation.expand(*batch_shape, right_permutation.size(-1))
    rows = torch.gather(dense, -2, lp.unsqueeze(-1).expand(*batch_shape, lp.size(-1), dense.size(-1)))
    return torch.gather(rows, -1, rp.unsqueeze(-2).expand(*batch_shape, lp.size(-1), rp.size(-1)))


def inverse_permutation(permutation):
    arange = torch.arange(permutation.size(-1), device=permutation.device)
    return torch.zeros_like(permutation).scatter_(-1, permutation, arange.expand_as(permutation))
