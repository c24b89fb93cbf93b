"""tools/vendor_int8_reference.py — what the vendor's own INT8 GEMM (hipBLASLt behind torch._int_mm) sustains on this GPU,
with full-entropy and with low-toggle operands, plus bf16 for scale.  A reference point for DESIGN.md §4.2: the fused
slice GEMM is compared with the datasheet peak in bench.py, but no kernel reaches that peak on real data."""
import torch, time
n = 8192
for ent in ("full", "low"):
    if ent == "full":
        a = torch.randint(-127, 128, (n, n), dtype=torch.int8, device="cuda")
        b = torch.randint(-127, 128, (n, n), dtype=torch.int8, device="cuda")
    else:
        a = torch.randint(0, 2, (n, n), dtype=torch.int8, device="cuda")
        b = torch.randint(0, 2, (n, n), dtype=torch.int8, device="cuda")
    for name, bb in (("NN", b), ("NT", b.t())):
        try:
            c = torch._int_mm(a, bb)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                c = torch._int_mm(a, bb)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            print(f"torch._int_mm {ent}-entropy {name} {n}^3: {dt*1e3:.3f} ms = {2*n**3/dt/1e12:.0f} TOPS", flush=True)
        except Exception as e:
            print("int_mm failed:", name, repr(e)[:200])
# fp8 / bf16 reference points
for dt_name, dt in (("bf16", torch.bfloat16),):
    x = torch.randn(n, n, device="cuda", dtype=dt); y = torch.randn(n, n, device="cuda", dtype=dt)
    z = x @ y; torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): z = x @ y
    torch.cuda.synchronize(); d = (time.perf_counter() - t0) / 20
    print(f"torch {dt_name} matmul {n}^3: {d*1e3:.3f} ms = {2*n**3/d/1e12:.0f} TFLOP/s")
