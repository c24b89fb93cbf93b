set -u
mkdir -p gpurun_out/r5e
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_forced_kernels.py tests/test_gpu_wide_kernel.py tests/test_gpu_parity.py -m gpu -q --maxfail=20 2>&1 | tail -12 > gpurun_out/r5e/gpu_tests_subset_tail.txt
tail -4 gpurun_out/r5e/gpu_tests_subset_tail.txt
timeout 900 python tools/ab.py --shapes 4096 8192 2048 4096x4096x1024 16384x16384x2048 3000x5000x4096 --modes fp64_int8_12 --variants auto k64 x16 wide rocblas --legs 5 > gpurun_out/r5e/s12_k64_ab.txt 2>&1
cat gpurun_out/r5e/s12_k64_ab.txt
