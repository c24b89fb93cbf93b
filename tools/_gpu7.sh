set -u
mkdir -p gpurun_out/r5g
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 2>&1 | tail -40 > gpurun_out/r5g/gpu_tests_tail.txt
tail -4 gpurun_out/r5g/gpu_tests_tail.txt
timeout 900 python bench.py > gpurun_out/r5g/bench.json 2> gpurun_out/r5g/bench.err
tail -c 400 gpurun_out/r5g/bench.json
bash tools/profile.sh r5g > gpurun_out/r5g/profile.log 2>&1
timeout 600 python tools/ab.py --preset short_k --variants auto classic wide k64 OZIMMU_HIP_GEMM_KERNEL=k64,OZIMMU_HIP_K64_BREG=1 rocblas --legs 5 > gpurun_out/r5g/short_k_policy_vs_forced.txt 2>&1
timeout 600 python tools/ab.py --preset mid --variants auto rocblas --legs 5 > gpurun_out/r5g/mid_sizes_vs_rocblas.txt 2>&1
tail -30 gpurun_out/r5g/short_k_policy_vs_forced.txt | cut -c1-260
