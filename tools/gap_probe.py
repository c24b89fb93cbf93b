"""tools/gap_probe.py M N K opa opb — time of one call as a function of the idle gap in front of it: calls back to back,
with a device synchronisation between calls, and with a synchronisation plus a host sleep; fp64_int8_9 with the 32x32x32
and the k64 tile function, and rocBLAS DGEMM.  (A kernel that starts on an idle part runs through the power manager's
start-of-load transient.)"""
import os; os.environ.setdefault("OZIMMU_HIP_ENV_PER_CALL", "1")
import sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ozimmu_amd as oz
m, n, k = (int(x) for x in sys.argv[1:4]); oa, ob = sys.argv[4:6]
h = oz.create(); oz.set_cuda_stream(h, torch.cuda.current_stream())
a = torch.rand((k, m) if oa == "N" else (m, k), dtype=torch.float64, device="cuda") * 2 - 1
b = torch.rand((n, k) if ob == "N" else (k, n), dtype=torch.float64, device="cuda") * 2 - 1
c = torch.zeros(n, m, dtype=torch.float64, device="cuda")
lda, ldb = a.shape[1], b.shape[1]
def ours(): oz.gemm(h, oa, ob, m, n, k, 1.0, a, lda, b, ldb, 0.0, c, m, "fp64_int8_9")
def vendor(): oz.native_dgemm(h, oa, ob, m, n, k, 1.0, a, lda, b, ldb, 0.0, c, m)
for name, fn, val in (("32x32x32 tile", ours, "0"), ("k64 tile", ours, "1"), ("rocBLAS DGEMM", vendor, "0")):
    os.environ["OZIMMU_HIP_K64_TILE"] = val
    row = []
    for gap in (None, 0.0, 0.001, 0.01, 0.1):
        for _ in range(4): fn()
        torch.cuda.synchronize()
        ts = []
        for i in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            if gap is not None:
                torch.cuda.synchronize()
                if gap: time.sleep(gap)
            ts.append((e0, e1))
        torch.cuda.synchronize()
        ms = sorted(x.elapsed_time(y) for x, y in ts[2:])
        row.append(f"{'back to back' if gap is None else f'sync+{gap * 1e3:g} ms'}: {ms[len(ms) // 2]:.2f}")
    print(f"{m}x{n}x{k} {name:14s} median ms per call | " + " | ".join(row), flush=True)
oz.destroy(h)
