import torch, sys
sys.path.insert(0, "/root/repo")
import ozimmu_amd as oz
m, n, k = (int(x) for x in sys.argv[1:4])
opb = sys.argv[4] if len(sys.argv) > 4 else "T"
beta = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
h = oz.create(); oz.set_cuda_stream(h, torch.cuda.current_stream())
a = torch.rand(k, m, dtype=torch.float64, device="cuda") * 2 - 1
b = torch.rand(k * n, dtype=torch.float64, device="cuda") * 2 - 1
c = torch.zeros(n, m, dtype=torch.float64, device="cuda")
for _ in range(10): oz.gemm(h, "N", opb, m, n, k, -1.0, a, m, b, n if opb == "T" else k, beta, c, m, "fp64_int8_9")
torch.cuda.synchronize()
for _ in range(3): oz.native_dgemm(h, "N", opb, m, n, k, -1.0, a, m, b, n if opb == "T" else k, beta, c, m)
torch.cuda.synchronize()
