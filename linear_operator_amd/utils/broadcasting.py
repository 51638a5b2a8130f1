"""Shape helper (reference: linear_operator/utils/broadcasting.py:7 `_matmul_broadcast_shape`)."""
from __future__ import annotations

import torch


def _matmul_broadcast_shape(shape_a, shape_b, error_msg=None):
    """Shape of a @ b with batch broadcasting; `shape_b` may be a vector."""
    m, n = shape_a[-2:]
    if len(shape_b) == 1:
        if n != shape_b[-1]:
            raise RuntimeError(error_msg or f"Incompatible dimensions for matmul: {tuple(shape_a)} and {tuple(shape_b)}")
        return torch.Size(tuple(shape_a[:-1]))
    if n != shape_b[-2]:
        raise RuntimeError(error_msg or f"Incompatible dimensions for matmul: {tuple(shape_a)} and {tuple(shape_b)}")
    try:
        batch = torch.broadcast_shapes(tuple(shape_a[:-2]), tuple(shape_b[:-2]))
    except RuntimeError:
        raise RuntimeError(error_msg or f"Incompatible batch dimensions for matmul: {tuple(shape_a)} and {tuple(shape_b)}")
    return torch.Size((*batch, m, shape_b[-1]))
