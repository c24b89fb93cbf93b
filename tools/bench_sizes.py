import torch, time, sys
sys.path.insert(0, "/root/repo")
import ozimmu_amd as oz
h = oz.create()
oz.set_cuda_stream(h, torch.cuda.current_stream())
for n in (1024, 1536, 2048, 3072, 4096, 6144, 8192):
    a = torch.rand(n, n, dtype=torch.float64, device="cuda") * 2 - 1
    b = torch.rand(n, n, dtype=torch.float64, device="cuda") * 2 - 1
    c = torch.zeros(n, n, dtype=torch.float64, device="cuda")
    def call(mode="fp64_int8_9"):
        if mode == "dgemm":
            oz.native_dgemm(h, "N", "N", n, n, n, 1.0, a, n, b, n, 0.0, c, n)
        else:
            oz.gemm(h, "N", "N", n, n, n, 1.0, a, n, b, n, 0.0, c, n, mode)
    res = {}
    for mode in ("fp64_int8_9", "dgemm"):
        call(mode); call(mode); torch.cuda.synchronize()
        reps = 50 if n <= 2048 else 10
        t0 = time.perf_counter()
        for _ in range(reps): call(mode)
        torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t0) / reps
    oz.enable_profiling(h)
    call(); torch.cuda.synchronize()
    st = oz.last_stage_ms(h)
    oz.disable_profiling(h)
    print(f"n={n}: ozaki {res['fp64_int8_9']*1e3:.3f} ms ({2*n**3/res['fp64_int8_9']/1e12:.1f} TF)  rocblas {res['dgemm']*1e3:.3f} ms ({2*n**3/res['dgemm']/1e12:.1f} TF)  stages split_A {st['split_A']:.3f} split_B {st['split_B']:.3f} gemm {st['int8tc']:.3f} ms", flush=True)
oz.destroy(h)
