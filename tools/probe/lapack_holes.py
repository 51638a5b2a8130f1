"""One batched LAPACK-style call of torch on the device per process (a failing one takes the context with it):
python lapack_holes.py <op> <batch> <n> <dtype>"""
import sys, torch
op, b, n, dt = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), getattr(torch, sys.argv[4])
dev = "cuda"
X = torch.randn(b, n, 24, device=dev, dtype=dt)
A = X @ X.mT + 0.5 * torch.eye(n, device=dev, dtype=dt)
try:
    if op == "eigh":
        ev, V = torch.linalg.eigh(A); torch.cuda.synchronize()
        err = float((V @ torch.diag_embed(ev) @ V.mT - A).abs().max())
    elif op == "solve":
        r = torch.randn(b, n, 1, device=dev, dtype=dt); x = torch.linalg.solve(A, r); torch.cuda.synchronize()
        err = float((A @ x - r).abs().max())
    elif op == "trsm":
        L = torch.linalg.cholesky(A[0]).expand(b, n, n).contiguous(); r = torch.randn(b, n, 15, device=dev, dtype=dt)
        x = torch.linalg.solve_triangular(L, r, upper=False); torch.cuda.synchronize()
        err = float((L @ x - r).abs().max())
    elif op == "qr":
        T = torch.randn(b, 20 * n, n, device=dev, dtype=dt); Q, R = torch.linalg.qr(T); torch.cuda.synchronize()
        err = float((Q @ R - T).abs().max())
    elif op == "logdet":
        ld = torch.logdet(A); torch.cuda.synchronize(); err = float((ld - torch.logdet(A.double().cpu()).to(dev)).abs().max())
    elif op == "inv":
        Ai = torch.linalg.inv(A); torch.cuda.synchronize(); err = float((Ai @ A - torch.eye(n, device=dev, dtype=dt)).abs().max())
    print(op, b, n, dt, "ok", f"{err:.2e}")
except Exception as e:  # noqa: BLE001
    print(op, b, n, dt, "FAILED", str(e).splitlines()[0])
