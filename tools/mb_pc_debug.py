"""Phase timers of the resident pivoted Cholesky: run with LO_OC_DEBUG=1 (prints 100 MHz tick counts of member 0)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linear_operator_amd import kernels as K
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(3)
B, N, R = 512, 8192, 32
Cm = torch.randn(B, N, R, generator=g, device=dev) / R ** 0.5
desc = K.lowrank_diag_descriptor(Cm, None)
for _ in range(3): K.pivoted_cholesky(desc, 15, contiguous=False)
