"""Throughput of the BASELINE configs beyond the headline one, wide vs classic slice-GEMM kernel (environment switch read
per call): C3 = fp64_int8_{3..18} at 4096^3, C5 = fp64_int8_9 32768 x 32768 x 1024 N/T, 16384^3 at S = 9 / 11 (C4's mode)."""
import os; os.environ.setdefault("OZIMMU_HIP_ENV_PER_CALL", "1")  # switches are flipped between calls (csrc/config.h)
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ozimmu_amd as oz
h = oz.create(); oz.set_cuda_stream(h, torch.cuda.current_stream())
def bench(m, n, k, opa, opb, mode, kernel, reps):
    if kernel: os.environ["OZIMMU_HIP_GEMM_KERNEL"] = kernel
    else: os.environ.pop("OZIMMU_HIP_GEMM_KERNEL", None)
    a = torch.rand((k, m) if opa == "N" else (m, k), dtype=torch.float64, device="cuda") * 2 - 1
    b = torch.rand((n, k) if opb == "N" else (k, n), dtype=torch.float64, device="cuda") * 2 - 1
    c = torch.zeros(n, m, dtype=torch.float64, device="cuda")
    lda, ldb = a.shape[1], b.shape[1]
    def call():
        if mode == "dgemm": oz.native_dgemm(h, opa, opb, m, n, k, 1.0, a, lda, b, ldb, 0.0, c, m)
        else: assert oz.gemm(h, opa, opb, m, n, k, 1.0, a, lda, b, ldb, 0.0, c, m, mode) == 0
    call(); call(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): call()
    torch.cuda.synchronize()
    return 2.0 * m * n * k * reps / (time.perf_counter() - t0) / 1e12
which = sys.argv[1:] or ["c3", "c5", "c4"]
if "c3" in which:
    n = 4096
    print(f"C3 {n}^3: rocBLAS DGEMM {bench(n, n, n, 'N', 'N', 'dgemm', None, 10):.1f} TF")
    for S in range(3, 19):
        w = bench(n, n, n, "N", "N", f"fp64_int8_{S}", "wide", 10); c = bench(n, n, n, "N", "N", f"fp64_int8_{S}", "classic", 10)
        d = bench(n, n, n, "N", "N", f"fp64_int8_{S}", None, 10)
        print(f"  fp64_int8_{S}: wide {w:6.1f}  classic {c:6.1f}  default {d:6.1f} TF", flush=True)
if "c5" in which:
    m = n = 32768; k = 1024
    print(f"C5 {m}x{n}x{k} N/T: rocBLAS {bench(m, n, k, 'N', 'T', 'dgemm', None, 5):.1f} TF  wide {bench(m, n, k, 'N', 'T', 'fp64_int8_9', 'wide', 5):.1f}"
          f"  classic {bench(m, n, k, 'N', 'T', 'fp64_int8_9', 'classic', 5):.1f} TF", flush=True)
if "c4" in which:
    n = 16384
    print(f"{n}^3: rocBLAS {bench(n, n, n, 'N', 'N', 'dgemm', None, 2):.1f} TF")
    for S in (8, 9, 11):
        print(f"  fp64_int8_{S}: wide {bench(n, n, n, 'N', 'N', f'fp64_int8_{S}', 'wide', 2):.1f}  classic {bench(n, n, n, 'N', 'N', f'fp64_int8_{S}', 'classic', 2):.1f} TF", flush=True)
oz.destroy(h)
