import torch, sys
sys.path.insert(0, "/root/repo")
import ozimmu_amd as oz
b, n = int(sys.argv[1]), int(sys.argv[2])
h = oz.create(); st = torch.cuda.current_stream(); oz.set_cuda_stream(h, st)
x = torch.rand(b, n, n, dtype=torch.float64, device="cuda") * 2 - 1
y = torch.rand(b, n, n, dtype=torch.float64, device="cuda") * 2 - 1
c = torch.zeros(b, n, n, dtype=torch.float64, device="cuda")
for _ in range(10):
    assert oz.gemm_strided_batched(h, st, "N", "N", n, n, n, 1.0, x, n, n * n, y, n, n * n, 0.0, c, n, n * n, b, "fp64_int8_9") == 0
torch.cuda.synchronize()
