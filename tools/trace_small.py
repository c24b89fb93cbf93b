import torch, sys
sys.path.insert(0, "/root/repo")
import ozimmu_amd as oz
n = int(sys.argv[1])
h = oz.create(); oz.set_cuda_stream(h, torch.cuda.current_stream())
a = torch.rand(n, n, dtype=torch.float64, device="cuda") * 2 - 1
b = torch.rand(n, n, dtype=torch.float64, device="cuda") * 2 - 1
c = torch.zeros(n, n, dtype=torch.float64, device="cuda")
for _ in range(20): oz.gemm(h, "N", "N", n, n, n, 1.0, a, n, b, n, 0.0, c, n, "fp64_int8_9")
torch.cuda.synchronize()
for _ in range(5): oz.native_dgemm(h, "N", "N", n, n, n, 1.0, a, n, b, n, 0.0, c, n)
torch.cuda.synchronize()
