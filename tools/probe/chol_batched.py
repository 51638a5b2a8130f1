import sys, torch
dev = "cuda"
b, n, dt = int(sys.argv[1]), int(sys.argv[2]), getattr(torch, sys.argv[3])
X = torch.randn(b, n, 20, device=dev, dtype=dt)
A = X @ X.mT + 0.5 * torch.eye(n, device=dev, dtype=dt)
try:
    L, info = torch.linalg.cholesky_ex(A)
    torch.cuda.synchronize()
    print(b, n, dt, "ok", float((L @ L.mT - A).abs().max()))
except Exception as e:
    print(b, n, dt, "FAILED", str(e).splitlines()[0])
