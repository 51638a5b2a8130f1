"""linear_operator_amd -- MI355X-native iterative solve / logdet hot path of cornellius-gp/linear_operator."""
