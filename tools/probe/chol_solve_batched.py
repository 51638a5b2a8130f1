import sys, torch
dev = "cuda"
b, n, c = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dt = getattr(torch, sys.argv[4]) if len(sys.argv) > 4 else torch.float32
X = torch.randn(b, n, 20, device=dev, dtype=dt)
A = X @ X.mT + 0.5 * torch.eye(n, device=dev, dtype=dt)
rhs = torch.randn(b, n, c, device=dev, dtype=dt)
try:
    L, info = torch.linalg.cholesky_ex(A); torch.cuda.synchronize(); print(b, n, c, "potrf ok", end="; ")
    x = torch.cholesky_solve(rhs, L); torch.cuda.synchronize()
    print("potrs ok", float((A @ x - rhs).abs().max()))
except Exception as e:
    print(b, n, c, "FAILED", str(e).splitlines()[0])
