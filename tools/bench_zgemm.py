"""ZGEMM (four real slice products) throughput by size, fp64_int8_9, 8 n^3 flops."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import ozimmu_amd as oz
h = oz.create(); oz.set_cuda_stream(h, torch.cuda.current_stream())
for n in [int(x) for x in sys.argv[1:]] or [1024, 2048, 4096, 8192]:
    a = torch.randn(n, n, dtype=torch.complex128, device="cuda")
    b = torch.randn(n, n, dtype=torch.complex128, device="cuda")
    c = torch.zeros(n, n, dtype=torch.complex128, device="cuda")
    def call(): assert oz.gemm(h, "N", "N", n, n, n, 1.0, a, n, b, n, 0.0, c, n, "fp64_int8_9", oz.complx) == 0
    reps = 20 if n <= 2048 else 5
    for _ in range(3): call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): call()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / reps
    print(f"zgemm n={n}: {t * 1e6:9.1f} us  {8 * n**3 / t / 1e12:6.1f} TF")
