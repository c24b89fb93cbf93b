"""Per-call time of torch.mm / torch.bmm float64 through LD_PRELOAD vs the same product through the direct API."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    n = int(sys.argv[2]); b = int(sys.argv[3])
    x = torch.rand(b, n, n, dtype=torch.float64, device="cuda") * 2 - 1
    y = torch.rand(b, n, n, dtype=torch.float64, device="cuda") * 2 - 1
    out = torch.empty(b, n, n, dtype=torch.float64, device="cuda")
    f = (lambda: torch.bmm(x, y, out=out)) if b > 1 else (lambda: torch.mm(x[0], y[0], out=out[0]))
    for _ in range(10): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): f()
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / 200 * 1e6:.1f}")
    sys.exit(0)
import torch
import ozimmu_amd as oz
lib = os.path.join(ROOT, "ozimmu_amd", "libozimmu_hip.so")
for (n, b) in ((1024, 1), (2048, 1), (1024, 8)):
    h = oz.create(); st = torch.cuda.current_stream(); oz.set_cuda_stream(h, st)
    x = torch.rand(b, n, n, dtype=torch.float64, device="cuda") * 2 - 1
    y = torch.rand(b, n, n, dtype=torch.float64, device="cuda") * 2 - 1
    c = torch.zeros(b, n, n, dtype=torch.float64, device="cuda")
    def call():
        if b == 1: assert oz.gemm(h, "N", "N", n, n, n, 1.0, x, n, y, n, 0.0, c, n, "fp64_int8_9") == 0
        else: assert oz.gemm_strided_batched(h, st, "N", "N", n, n, n, 1.0, x, n, n * n, y, n, n * n, 0.0, c, n, n * n, b, "fp64_int8_9") == 0
    for _ in range(10): call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): call()
    torch.cuda.synchronize(); direct = (time.perf_counter() - t0) / 200 * 1e6
    oz.destroy(h)
    res = {}
    for name, extra in (("native", {}), ("preload", dict(LD_PRELOAD=lib, OZIMMU_COMPUTE_MODE="fp64_int8_9"))):
        e = {k: v for k, v in os.environ.items() if not k.startswith("OZIMMU_")}; e.update(extra)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n), str(b)], env=e, capture_output=True, text=True).stdout.strip().splitlines()
        res[name] = out[-1] if out else "?"
    print(f"n={n} batch={b}: direct API {direct:.1f} us | torch through preload {res['preload']} us | torch native rocBLAS {res['native']} us", flush=True)
