"""The operator classes that feed the iterative solve / logdet path."""
from ._linear_operator import LinearOperator, to_dense
from .added_diag_linear_operator import AddedDiagLinearOperator
from .dense_linear_operator import DenseLinearOperator, to_linear_operator
from .diag_linear_operator import ConstantDiagLinearOperator, DiagLinearOperator
from .identity_linear_operator import IdentityLinearOperator
from .kronecker_product_linear_operator import KroneckerProductDiagLinearOperator, KroneckerProductLinearOperator
from .kronecker_product_added_diag_linear_operator import KroneckerProductAddedDiagLinearOperator
from .linear_operator_representation_tree import LinearOperatorRepresentationTree
from .low_rank_root_added_diag_linear_operator import LowRankRootAddedDiagLinearOperator
from .matmul_linear_operator import MatmulLinearOperator
from .root_linear_operator import LowRankRootLinearOperator, RootLinearOperator
from .sum_linear_operator import PsdSumLinearOperator, SumLinearOperator
from .triangular_linear_operator import TriangularLinearOperator

__all__ = [
    "LowRankRootAddedDiagLinearOperator", "KroneckerProductAddedDiagLinearOperator",
    "LinearOperator", "to_dense", "to_linear_operator", "AddedDiagLinearOperator", "DenseLinearOperator",
    "DiagLinearOperator", "ConstantDiagLinearOperator", "IdentityLinearOperator", "KroneckerProductLinearOperator", "KroneckerProductDiagLinearOperator",
    "LinearOperatorRepresentationTree", "RootLinearOperator", "LowRankRootLinearOperator", "SumLinearOperator",
    "PsdSumLinearOperator", "TriangularLinearOperator", "MatmulLinearOperator",
]
