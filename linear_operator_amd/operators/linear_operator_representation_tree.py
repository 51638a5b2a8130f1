"""Flatten / rebuild an operator tree from its leaf tensors, so autograd Functions can take plain tensors
(reference: linear_operator/operators/linear_operator_representation_tree.py:8-44)."""
from __future__ import annotations

import itertools


class LinearOperatorRepresentationTree(object):
    def __init__(self, linear_op):
        self._cls = linear_op.__class__
        self._kw_names = list(linear_op._differentiable_kwargs.keys())
        self._static_kwargs = linear_op._nondifferentiable_kwargs
        self.children = []  # (index or slice into the flat tensor list, subtree or None)
        pos = 0
        for arg in itertools.chain(linear_op._args, linear_op._differentiable_kwargs.values()):
            if hasattr(arg, "representation") and callable(arg.representation):
                n = len(arg.representation())
                self.children.append((slice(pos, pos + n), arg.representation_tree()))
                pos += n
            else:
                self.children.append((pos, None))
                pos += 1

    def __call__(self, *flat):
        rebuilt = [flat[idx] if sub is None else sub(*flat[idx]) for idx, sub in self.children]
        nkw = len(self._kw_names)
        if nkw:
            args, kwvals = rebuilt[:-nkw], rebuilt[-nkw:]
            return self._cls(*args, **dict(zip(self._kw_names, kwvals)), **self._static_kwargs)
        return self._cls(*rebuilt, **self._static_kwargs)
