from .root_linear_operator import LowRankRootLinearOperator  # noqa: F401  (reference module layout)
