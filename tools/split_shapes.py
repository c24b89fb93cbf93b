"""Split-stage time (row max + cut, stage events) for operand shapes with the same byte count."""
import sys, torch
sys.path.insert(0, "/root/repo")
import ozimmu_amd as oz
h = oz.create(); st = torch.cuda.current_stream(); oz.set_cuda_stream(h, st)
oz.enable_profiling(h)
def run(m, n, k, opa="N", opb="N", batch=1):
    a = torch.rand(batch, m * k, dtype=torch.float64, device="cuda") * 2 - 1
    b = torch.rand(batch, n * k, dtype=torch.float64, device="cuda") * 2 - 1
    c = torch.zeros(batch, m * n, dtype=torch.float64, device="cuda")
    lda = m if opa == "N" else k
    ldb = k if opb == "N" else n
    ts = []
    for i in range(10):
        if batch == 1:
            assert oz.gemm(h, opa, opb, m, n, k, 1.0, a, lda, b, ldb, 0.0, c, m, "fp64_int8_9") == 0
        else:
            assert oz.gemm_strided_batched(h, st, opa, opb, m, n, k, 1.0, a, lda, m * k, b, ldb, n * k, 0.0, c, m, m * n, batch, "fp64_int8_9") == 0
        x = oz.last_stage_ms(h)
        if i >= 3: ts.append(((x["split_A"] + x["split_B"]) * 1e3, x["int8tc"] * 1e3))
    ts.sort()
    s, g = ts[len(ts) // 2]
    mb = 8 * (m + n) * k * batch / 1e6
    print(f"{opa}{opb} m={m} n={n} k={k} batch={batch}: split {s:7.1f} us ({mb:6.1f} MB in, {mb * (8 + 9) / 8 / s / 1e3:5.2f} TB/s algorithmic)  gemm {g:8.1f} us")
run(2048, 2048, 2048)
run(1024, 1024, 1024, batch=8)
run(8192, 8192, 1024)
run(8192, 8192, 1024, "T", "N")
run(8192, 8192, 1024, "N", "T")
run(1024, 1024, 8192)
run(1024, 1024, 1024, "T", "N", batch=8)
run(1024, 1024, 1024, "N", "T", batch=8)
run(1024, 1024, 1024)
run(4096, 4096, 4096)
