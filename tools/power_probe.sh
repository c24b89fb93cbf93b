#!/bin/bash
# tools/power_probe.sh — sample socket power and shader clock (rocm-smi) while a GEMM loop runs; GPU box only.
# Evidence for DESIGN.md §4.2: the fused kernel runs at the board's power limit with the clock pulled down.
R=${GRAFT_REPO_ROOT:-$(pwd)}
rocm-smi --showmaxpower --showperflevel 2>/dev/null | grep -E "Max|Perf" | head -4
for what in "fp64_int8_9" "dgemm"; do
  python - "$what" <<PY &
import sys, time, torch
sys.path.insert(0, "$R")
import ozimmu_amd as oz
mode = sys.argv[1]
n = 8192
h = oz.create(); oz.set_cuda_stream(h, torch.cuda.current_stream())
a = torch.rand(n, n, dtype=torch.float64, device="cuda") * 2 - 1
b = torch.rand(n, n, dtype=torch.float64, device="cuda") * 2 - 1
c = torch.zeros(n, n, dtype=torch.float64, device="cuda")
t0 = time.time()
while time.time() - t0 < 8:
    for _ in range(10):
        if mode == "dgemm": oz.native_dgemm(h, "N", "N", n, n, n, 1.0, a, n, b, n, 0.0, c, n)
        else: oz.gemm(h, "N", "N", n, n, n, 1.0, a, n, b, n, 0.0, c, n, mode)
    torch.cuda.synchronize()
PY
  PID=$!
  sleep 4
  echo "== while running $what"
  for i in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr -s ' ' | head -4; sleep 0.7; done
  wait $PID
done
