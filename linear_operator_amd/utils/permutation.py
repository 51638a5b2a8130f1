"""apply_permutation / inverse_permutation (reference behaviour: linear_operator/utils/permutation.py:9-102).

`apply_permutation(K, left, right)` returns the entries `K[..., left[i], right[j]]` -- `Pi_left K Pi_right^T` for
(batched, possibly PARTIAL) permutation vectors.  For an operator the entries are fetched through its element access
(`__getitem__` with broadcast index tensors -> `_get_indices`, operators/_linear_operator.py): `len(left) x len(right)`
elements are generated from the structure (root rows, Kronecker factors, ...), the N x N matrix is never formed -- a
single pivot row of an operator of size 65536 costs 65536 elements, not 2^32.  Tensors are gathered directly.
The pivoted-Cholesky kernels generate their permuted rows on the device from the operator descriptor; these helpers
serve callers that post-process pivots and the generic row fetch of operators without a descriptor.
"""
from __future__ import annotations

import torch


def _index_grid(batch_shape, left, right, device):
    """Index tensors (one per batch dimension, rows, columns), each broadcastable to [*batch, len(left), len(right)]."""
    nb = len(batch_shape)
    batch_idx = []
    for dim, size in enumerate(batch_shape):
        view = [1] * (nb + 2)
        view[dim] = size
        batch_idx.append(torch.arange(size, device=device).view(view))
    rows = left.expand(*batch_shape, left.size(-1)).unsqueeze(-1)
    cols = right.expand(*batch_shape, right.size(-1)).unsqueeze(-2)
    return batch_idx, rows, cols


def apply_permutation(matrix, left_permutation=None, right_permutation=None):
    if left_permutation is None and right_permutation is None:
        return matrix
    device = matrix.device
    batch_shape = tuple(matrix.shape[:-2])
    # the batch of the result is the broadcast of the operator's batch and the permutations' leading dimensions
    for perm in (left_permutation, right_permutation):
        if perm is not None:
            batch_shape = tuple(torch.broadcast_shapes(batch_shape, perm.shape[:-1]))
    if left_permutation is None:
        left_permutation = torch.arange(matrix.size(-2), device=device)
    if right_permutation is None:
        right_permutation = torch.arange(matrix.size(-1), device=device)
    if tuple(matrix.shape[:-2]) != batch_shape:
        matrix = (matrix.expand(*batch_shape, *matrix.shape[-2:]) if torch.is_tensor(matrix)
                  else matrix._expand_batch(batch_shape))
    batch_idx, rows, cols = _index_grid(batch_shape, left_permutation, right_permutation, device)
    return matrix[(*batch_idx, rows, cols)]


def inverse_permutation(permutation):
    """The vector q with q[p[i]] = i, batched over the leading dimensions."""
    ranks = torch.arange(permutation.size(-1), device=permutation.device).expand_as(permutation)
    return torch.empty_like(permutation).scatter_(-1, permutation, ranks)
