"""Matmul Function (reference: linear_operator/functions/_matmul.py:9-66), forward only."""
from __future__ import annotations

import torch
from torch.autograd import Function

from ._common import not_yet


class Matmul(Function):
    @staticmethod
    def forward(ctx, representation_tree, rhs, *matrix_args):
        linear_op = representation_tree(*matrix_args)
        is_vec = rhs.ndimension() == 1
        if is_vec:
            rhs = rhs.unsqueeze(-1)
        res = linear_op._matmul(rhs)
        return res.squeeze(-1) if is_vec else res

    @staticmethod
    def backward(ctx, grad_output):
        not_yet("Matmul")
