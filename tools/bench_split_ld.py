import torch, sys
sys.path.insert(0, "/root/repo")
import ozimmu_amd as oz
h = oz.create()
oz.set_cuda_stream(h, torch.cuda.current_stream())
n = 8192
for pad in (0, 16, 32, 48, 272):
    ld = n + pad
    a = torch.rand(n, ld, dtype=torch.float64, device="cuda") * 2 - 1   # column-major n x n with leading dim ld
    b = torch.rand(n, ld, dtype=torch.float64, device="cuda") * 2 - 1
    c = torch.zeros(n, n, dtype=torch.float64, device="cuda")
    oz.enable_profiling(h)
    for ops in (("N", "N"), ("T", "T")):
        for _ in range(3):
            oz.gemm(h, ops[0], ops[1], n, n, n, 1.0, a, ld, b, ld, 0.0, c, n, "fp64_int8_9")
            torch.cuda.synchronize()
        st = oz.last_stage_ms(h)
        print(f"ld={ld} ops={ops}: split_A {st['split_A']:.3f} split_B {st['split_B']:.3f} gemm {st['int8tc']:.3f}", flush=True)
    oz.disable_profiling(h)
oz.destroy(h)
