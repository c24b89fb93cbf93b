"""tools/host_time.py — host time to enqueue one call (loop of 50 queued calls timed before the synchronisation) next to the
device time per call, fp64_int8_9 and rocBLAS DGEMM: what a caller that synchronises after every call pays on top."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ozimmu_amd as oz
h = oz.create(); oz.set_cuda_stream(h, torch.cuda.current_stream())
for (m, n, k, oa, ob) in [(256, 256, 256, "N", "N"), (1024, 1024, 1024, "N", "N"), (2048, 2048, 2048, "N", "N"), (4096, 4096, 4096, "N", "N"),
                          (8192, 8192, 8192, "N", "N"), (32768, 32768, 1024, "N", "T"), (16384, 16384, 16384, "N", "N")]:
    a = torch.rand((k, m) if oa == "N" else (m, k), dtype=torch.float64, device="cuda") * 2 - 1
    b = torch.rand((n, k) if ob == "N" else (k, n), dtype=torch.float64, device="cuda") * 2 - 1
    c = torch.zeros(n, m, dtype=torch.float64, device="cuda")
    lda, ldb = a.shape[1], b.shape[1]
    row = []
    for name, fn in (("fp64_int8_9", lambda: oz.gemm(h, oa, ob, m, n, k, 1.0, a, lda, b, ldb, 0.0, c, m, "fp64_int8_9")),
                     ("rocBLAS", lambda: oz.native_dgemm(h, oa, ob, m, n, k, 1.0, a, lda, b, ldb, 0.0, c, m))):
        reps = 50 if m * n * k < 2e11 else 6
        for _ in range(3): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        row.append(f"{name}: host {(t1 - t0) / reps * 1e6:8.1f} us / call, device {(t2 - t0) / reps * 1e6:9.1f} us / call")
    print(f"{m}x{n}x{k}: " + " | ".join(row), flush=True)
    del a, b, c
oz.destroy(h)
