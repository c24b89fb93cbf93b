set -u
mkdir -p gpurun_out/r5b
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --maxfail=30 2>&1 | tail -40 > gpurun_out/r5b/gpu_tests_tail.txt
tail -8 gpurun_out/r5b/gpu_tests_tail.txt
timeout 900 python tools/ab.py --shapes 4096 8192 2048 4096x4096x1024 16384x16384x2048 3000x5000x4096 --modes fp64_int8_11 --variants auto k64 x16 wide rocblas --legs 5 > gpurun_out/r5b/s11_k64_ab.txt 2>&1
cat gpurun_out/r5b/s11_k64_ab.txt
timeout 900 python bench.py > gpurun_out/r5b/bench.json 2> gpurun_out/r5b/bench.err
tail -c 1500 gpurun_out/r5b/bench.json
