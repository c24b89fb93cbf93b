#!/usr/bin/env python
"""tools/c5_calls.py [reps=6] [M N K opA opB mode] - a handful of calls of one shape (default: BASELINE C5, fp64_int8_9 32768 x 32768 x 1024
N/T) and nothing else, for `rocprofv3 --kernel-trace --stats -- python tools/c5_calls.py`: which kernels a call of that shape is made of
and what each costs (profiles/r6_c5_kernel_stats.csv).  Prints the whole-call time next to the vendor DGEMM's."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ozimmu_amd as oz

a = sys.argv[1:]
reps = int(a[0]) if a else 6
m, n, k = (int(x) for x in a[1:4]) if len(a) >= 4 else (32768, 32768, 1024)
opa, opb = (a[4], a[5]) if len(a) >= 6 else ("N", "T")
mode = a[6] if len(a) >= 7 else "fp64_int8_9"
h = oz.create()
oz.set_cuda_stream(h, torch.cuda.current_stream())
A = torch.rand((k, m) if opa == "N" else (m, k), dtype=torch.float64, device="cuda") * 2 - 1
B = torch.rand((n, k) if opb == "N" else (k, n), dtype=torch.float64, device="cuda") * 2 - 1
C = torch.zeros(n, m, dtype=torch.float64, device="cuda")
lda, ldb = A.shape[1], B.shape[1]


def timed(call):
    call(); call(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def ours():
    assert oz.gemm(h, opa, opb, m, n, k, 1.0, A, lda, B, ldb, 0.0, C, m, mode) == 0


t = timed(ours)
tv = timed(lambda: oz.native_dgemm(h, opa, opb, m, n, k, 1.0, A, lda, B, ldb, 0.0, C, m))
fl = 2.0 * m * n * k
print(f"{m}x{n}x{k} {opa}/{opb} {mode}: {t * 1e3:.3f} ms = {fl / t / 1e12:.2f} TFLOP/s ({oz.last_kernel(h)[0]}); vendor DGEMM {tv * 1e3:.3f} ms = {fl / tv / 1e12:.2f}; "
      f"ratio {tv / t:.3f}")
oz.destroy(h)
