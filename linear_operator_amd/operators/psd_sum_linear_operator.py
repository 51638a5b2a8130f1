from .sum_linear_operator import PsdSumLinearOperator  # noqa: F401  (reference module layout)
