"""k_kron_fused (both GEMMs of the Kronecker matvec in one launch, the intermediate in the accumulators) against the
two-launch path (LO_NO_KRON_FUSED=1) and a float64 reference."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import _hip, kernels as K
dev = torch.device("cuda")
for (B, n1, n2) in ((128, 256, 256), (128, 128, 128), (5, 256, 128), (3, 128, 384)):
    g = torch.Generator(device=dev); g.manual_seed(B + n1)
    K1 = torch.randn(B, n1, n1, generator=g, device=dev) / n1 ** 0.5; K1 = K1 + K1.mT
    K2 = torch.randn(B, n2, n2, generator=g, device=dev) / n2 ** 0.5; K2 = K2 + K2.mT
    sig = torch.rand(B, generator=g, device=dev) + 0.5
    v = torch.randn(B, n1 * n2, 1, generator=g, device=dev)
    desc = K.kron_diag_descriptor(K1, K2, sig, const_diag=True)
    os.environ.pop("LO_NO_KRON_FUSED", None)
    y1 = K.matvec(desc, v)
    os.environ["LO_NO_KRON_FUSED"] = "1"
    y0 = K.matvec(desc, v)
    os.environ.pop("LO_NO_KRON_FUSED")
    V = v.reshape(B, n1, n2).double()
    ref = (K1.double() @ V @ K2.double().mT).reshape(B, -1, 1) + sig.double()[:, None, None] * v.double()
    e1 = float(((y1.double() - ref).norm(dim=-2) / ref.norm(dim=-2)).max())
    e0 = float(((y0.double() - ref).norm(dim=-2) / ref.norm(dim=-2)).max())
    def t(fn, reps=20):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
    tf = t(lambda: K.matvec(desc, v))
    os.environ["LO_NO_KRON_FUSED"] = "1"
    t2 = t(lambda: K.matvec(desc, v))
    os.environ.pop("LO_NO_KRON_FUSED")
    flop = 2.0 * B * n1 * n2 * (n1 + n2)
    print(f"B={B} {n1}x{n2}: rel err vs fp64 fused {e1:.2e} two-launch {e0:.2e}; fused {tf:.1f} us ({flop / tf / 1e6:.1f} TFLOP/s), "
          f"two launches {t2:.1f} us ({flop / t2 / 1e6:.1f} TFLOP/s)")
