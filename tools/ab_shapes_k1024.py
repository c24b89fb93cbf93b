import sys, time, torch
sys.path.insert(0, "/root/repo")
import ozimmu_amd as oz
h = oz.create(); oz.set_cuda_stream(h, torch.cuda.current_stream())
for (m, n, k, reps) in [(16384, 16384, 1024, 10), (8192, 8192, 1024, 30), (8192, 8192, 1536, 20), (4096, 4096, 1024, 60)]:
    a = torch.rand(k, m, dtype=torch.float64, device="cuda") * 2 - 1
    b = torch.rand(n, k, dtype=torch.float64, device="cuda") * 2 - 1
    c = torch.zeros(n, m, dtype=torch.float64, device="cuda")
    def call(): assert oz.gemm(h, "N", "N", m, n, k, 1.0, a, m, b, k, 0.0, c, m, "fp64_int8_9") == 0
    ts = []
    for r in range(3):
        call(); call(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): call()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / reps)
    t = sorted(ts)[1]
    print(f"{m}x{n}x{k}: {t * 1e6:9.1f} us  {2 * m * n * k / t / 1e12:6.1f} TF", flush=True)
