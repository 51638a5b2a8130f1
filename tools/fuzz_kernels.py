"""The small kernels around the solvers on random shapes, each against float64 formulas: preconditioner build + apply
(ranks 1 .. 128, constant / full diagonal, both factor layouts), the bilinear derivatives (dense, diag, root, Kronecker),
the SLQ eigen-solver of the CG tridiagonals.  `python tools/fuzz_kernels.py --minutes 3 --seed 1` on the GPU box."""
import argparse, os, random, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from linear_operator_amd import kernels as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=3.0)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
rnd = random.Random(args.seed)
dev = torch.device("cuda")
t_end = time.time() + 60 * args.minutes
counts = {}


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


while time.time() < t_end:
    g = torch.Generator(device=dev); g.manual_seed(rnd.randrange(1 << 30))
    what = rnd.choice(["precond", "bil_dense", "bil_diag", "bil_root", "bil_kron", "slq"])
    B = rnd.choice([1, 2, 7, 33])
    if what == "precond":
        N = rnd.choice([40, 257, 1000, 4099, 8192])
        k = min(rnd.choice([1, 2, 5, 15, 16, 17, 32, 33, 64, 100, 128]), N)
        const = rnd.random() < 0.4
        rows_layout = rnd.random() < 0.5  # the [B, k, N] rows the pivoted-Cholesky kernels write, as a strided view
        Lr = torch.randn(B, k, N, generator=g, device=dev) / k ** 0.5
        L = Lr.mT if rows_layout else Lr.mT.contiguous()
        d = (torch.rand(B, generator=g, device=dev) + 0.3) if const else (torch.rand(B, N, generator=g, device=dev) + 0.3)
        c = rnd.choice([1, 3, 17])
        r = torch.randn(B, N, c, generator=g, device=dev)
        tag = (what, B, N, k, const, rows_layout, c)
        if os.environ.get("FUZZ_VERBOSE"):
            print(tag, flush=True)
        pre = K.precond_build(L, d, const)
        z = K.precond_apply(pre, r)
        dd = (d.double()[:, None].expand(B, N) if const else d.double())
        P64 = L.double() @ L.double().mT + torch.diag_embed(dd)
        ex = torch.linalg.solve(P64, r.double())
        assert rel(z, ex) < 2e-4, (tag, rel(z, ex))
        assert rel(pre.logdet, torch.logdet(P64)) < 1e-5, (tag, "logdet")
    elif what in ("bil_dense", "bil_diag", "bil_root", "bil_kron"):
        D = rnd.choice([1, 3, 17, 34, 64])
        if what == "bil_kron":
            n1, n2 = rnd.choice([(8, 12), (33, 40), (64, 64)])
            N = n1 * n2
            B = min(B, 7)
        else:
            N = rnd.choice([30, 257, 1000, 3001])
        U = torch.randn(B, N, D, generator=g, device=dev)
        V = torch.randn(B, N, D, generator=g, device=dev)
        tag = (what, B, N, D)
        if os.environ.get("FUZZ_VERBOSE"):
            print(tag, flush=True)
        if what == "bil_dense":
            if B * N * N > (1 << 28):
                continue
            out = K.bilinear_dense(U, V, (B,))
            assert rel(out, U.double() @ V.double().mT) < 1e-5, tag
        elif what == "bil_diag":
            const = rnd.random() < 0.5
            out = K.bilinear_diag(U, V, (B,), constant=const)
            ex = (U.double() * V.double()).sum(-1)
            ex = ex.sum(-1, keepdim=True) if const else ex
            scale = (U.double() * V.double()).abs().sum((-2, -1), keepdim=False).max().item()  # (a sum of signed terms)
            assert out.shape == ex.shape and (out.double() - ex).abs().max().item() < 1e-6 * scale, (tag, const, tuple(out.shape))
        elif what == "bil_root":
            R = rnd.choice([1, 8, 20, 32, 48])
            Cm = torch.randn(B, N, R, generator=g, device=dev)
            out, rowdot = K.bilinear_root(Cm, U, V, with_rowdot=True)
            ex = U.double() @ (V.double().mT @ Cm.double()) + V.double() @ (U.double().mT @ Cm.double())
            assert rel(out, ex) < 1e-5 and rel(rowdot, (U.double() * V.double()).sum(-1)) < 1e-5, (tag, R)
        else:
            K1 = torch.randn(B, n1, n1, generator=g, device=dev); K2 = torch.randn(B, n2, n2, generator=g, device=dev)
            d1, d2 = K.bilinear_kron(K1, K2, U, V)
            U4, V4 = U.double().reshape(B, n1, n2, D), V.double().reshape(B, n1, n2, D)
            e1 = torch.einsum("bikd,bkl,bjld->bij", U4, K2.double(), V4)
            e2 = torch.einsum("bikd,bij,bjld->bkl", U4, K1.double(), V4)
            assert rel(d1, e1) < 1e-5 and rel(d2, e2) < 1e-5, tag
    else:
        P, T = rnd.choice([1, 10, 16, 33]), rnd.choice([1, 2, 7, 20, 32])
        al = torch.rand(P, B, T, generator=g, device=dev) + 1.0
        be = torch.rand(P, B, max(T - 1, 0), generator=g, device=dev) * 0.4
        t = torch.diag_embed(al)
        if T > 1:
            t = t + torch.diag_embed(be, offset=1) + torch.diag_embed(be, offset=-1)
        n = rnd.choice([100, 8192])
        tag = (what, P, B, T)
        if os.environ.get("FUZZ_VERBOSE"):
            print(tag, flush=True)
        ev, evec, ld = K.tridiag_eigh_slq(t.contiguous(), n, want_evecs=True)
        e64, v64 = torch.linalg.eigh(t.double())
        assert rel(ev.sort(-1).values, e64) < 1e-5, tag
        ex = (v64[..., 0, :] ** 2 * e64.log()).sum(-1).mean(0) * n  # stochastic_lq.py: first-row weights, mean over probes
        assert rel(ld, ex) < 1e-4, (tag, rel(ld, ex))
    counts[what] = counts.get(what, 0) + 1
print("fuzz ok:", counts, "seed", args.seed)
