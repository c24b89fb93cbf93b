import torch, time
torch.manual_seed(0)
n = 4096
a = torch.rand(n, n, dtype=torch.float64, device="cuda") + n * torch.eye(n, dtype=torch.float64, device="cuda")
b = torch.rand(n, 64, dtype=torch.float64, device="cuda")
for name, f in [("solve", lambda: torch.linalg.solve(a, b)),
                ("cholesky", lambda: torch.linalg.cholesky(a @ a.T + torch.eye(n, dtype=torch.float64, device="cuda"))),
                ("lu_factor", lambda: torch.linalg.lu_factor(a)[0]),
                ("qr", lambda: torch.linalg.qr(a)[1]),
                ("matmul chain", lambda: (a @ a) @ b),
                ("addmm", lambda: torch.addmm(b, a, b, beta=0.5, alpha=2.0)),
                ("baddbmm", lambda: torch.baddbmm(torch.ones(4, 1024, 1024, dtype=torch.float64, device="cuda"), a[:1024, :1024].expand(4, 1024, 1024).contiguous(), a[:1024, 1024:2048].expand(4, 1024, 1024).contiguous())),
                ("einsum", lambda: torch.einsum("ij,jk->ik", a, a.T.contiguous()))]:
    try:
        t0 = time.time(); r = f(); torch.cuda.synchronize()
        print(f"{name}: ok {time.time()-t0:.2f}s finite={bool(torch.isfinite(r).all())}")
    except Exception as e:
        print(f"{name}: FAILED {type(e).__name__}: {str(e)[:200]}")
x = torch.linalg.solve(a, b)
print("solve residual %.3e" % ((a @ x - b).abs().max() / b.abs().max()).item())
