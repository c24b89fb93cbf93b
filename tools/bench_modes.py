"""tools/bench_modes.py [n] [modes...] — DGEMM-equivalent TFLOP/s of the given modes at n^3 (min of 3 rounds)."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import ozimmu_amd as oz
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
modes = sys.argv[2:] or ["fp64_int8_9"]
h = oz.create(); oz.set_cuda_stream(h, torch.cuda.current_stream())
a = torch.rand(n, n, dtype=torch.float64, device="cuda") * 2 - 1
b = torch.rand(n, n, dtype=torch.float64, device="cuda") * 2 - 1
c = torch.zeros(n, n, dtype=torch.float64, device="cuda")
for mode in modes:
    best = 1e9
    for rnd in range(3):
        for _ in range(2): oz.gemm(h, "N", "N", n, n, n, 1.0, a, n, b, n, 0.0, c, n, mode)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): oz.gemm(h, "N", "N", n, n, n, 1.0, a, n, b, n, 0.0, c, n, mode)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 5)
    S = oz.get_num_split(mode)
    print(f"n={n} {mode}: {2*n**3/best/1e12:7.1f} TF  ({2*n**3/best/1e12*S*(S+1)/2:6.0f} INT8 TOPS)", flush=True)
oz.destroy(h)
