set -u
mkdir -p gpurun_out/r5d
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --maxfail=30 2>&1 | tail -40 > gpurun_out/r5d/gpu_tests_tail.txt
tail -6 gpurun_out/r5d/gpu_tests_tail.txt
timeout 700 python tools/ab.py --shapes 8192x8192x256 8192x8192x512 8192x8192x1024 4096x4096x1024 16384x16384x512 4096x4096x512 2048x2048x2048 8192 --modes fp64_int8_9 --variants OZIMMU_HIP_EPI_OVERLAP=0 OZIMMU_HIP_EPI_OVERLAP=1 rocblas --legs 7 > gpurun_out/r5d/epi_overlap_named_ab.txt 2>&1
timeout 300 python tools/ab.py --shapes 32768x32768x1024 --ops NT --modes fp64_int8_9 --variants OZIMMU_HIP_EPI_OVERLAP=0 OZIMMU_HIP_EPI_OVERLAP=1 rocblas --legs 5 >> gpurun_out/r5d/epi_overlap_named_ab.txt 2>&1
cat gpurun_out/r5d/epi_overlap_named_ab.txt
timeout 900 python bench.py > gpurun_out/r5d/bench.json 2> gpurun_out/r5d/bench.err
tail -c 600 gpurun_out/r5d/bench.json
