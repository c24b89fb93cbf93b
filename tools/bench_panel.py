"""Panel-update shapes (blocked factorisations: C -= A B with a short K): fp64_int8_9 vs native rocBLAS DGEMM."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import ozimmu_amd as oz
h = oz.create(); oz.set_cuda_stream(h, torch.cuda.current_stream())
for mn in (4096, 8192, 16384):
    for k in (128, 256, 512, 1024, 2048):
        a = torch.rand(k, mn, dtype=torch.float64, device="cuda") * 2 - 1   # m x k, column-major
        b = torch.rand(k, mn, dtype=torch.float64, device="cuda") * 2 - 1   # n x k stored column-major = op T
        c = torch.zeros(mn, mn, dtype=torch.float64, device="cuda")
        res = {}
        for mode in ("fp64_int8_9", "fp64_int8_8", "dgemm"):
            def call():
                if mode == "dgemm": oz.native_dgemm(h, "N", "T", mn, mn, k, -1.0, a, mn, b, mn, 1.0, c, mn)
                else: assert oz.gemm(h, "N", "T", mn, mn, k, -1.0, a, mn, b, mn, 1.0, c, mn, mode) == 0
            reps = max(3, int(4e11 / (mn * mn * k)))
            call(); call(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): call()
            torch.cuda.synchronize(); res[mode] = 2.0 * mn * mn * k * reps / (time.perf_counter() - t0) / 1e12
        print(f"m=n={mn} k={k}: fp64_int8_9 {res['fp64_int8_9']:6.1f}  fp64_int8_8 {res['fp64_int8_8']:6.1f}  rocBLAS {res['dgemm']:6.1f} TF  "
              f"({res['fp64_int8_9'] / res['dgemm']:.2f}x / {res['fp64_int8_8'] / res['dgemm']:.2f}x)", flush=True)
