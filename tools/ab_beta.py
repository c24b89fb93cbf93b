import sys, time, torch
sys.path.insert(0, "/root/repo")
import ozimmu_amd as oz
h = oz.create(); oz.set_cuda_stream(h, torch.cuda.current_stream())
for (mn, k) in ((8192, 1024), (8192, 2048), (8192, 8192), (4096, 4096), (2048, 2048)):
    a = torch.rand(k, mn, dtype=torch.float64, device="cuda") * 2 - 1
    b = torch.rand(k, mn, dtype=torch.float64, device="cuda") * 2 - 1
    c = torch.zeros(mn, mn, dtype=torch.float64, device="cuda")
    out = []
    for (opb, ldb) in (("T", mn), ("N", k)):
        for beta in (0.0, 1.0):
            def call(): assert oz.gemm(h, "N", opb, mn, mn, k, -1.0, a, mn, b, ldb, beta, c, mn, "fp64_int8_9") == 0
            reps = max(3, int(4e11 / (mn * mn * k)))
            best = 0
            for r in range(2):
                call(); call(); torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(reps): call()
                torch.cuda.synchronize(); best = max(best, 2.0 * mn * mn * k * reps / (time.perf_counter() - t0) / 1e12)
            out.append(f"N{opb} beta={beta:g}: {best:5.1f}")
    print(f"m=n={mn} k={k}: " + "  ".join(out), flush=True)
