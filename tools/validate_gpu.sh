# tools/validate_gpu.sh - one GPU-box call (gpurun -- bash tools/validate_gpu.sh): the whole GPU suite, bench.py, rocprofv3 kernel stats + PMC
# (tools/profile.sh -> gpurun_out/prof_validate, condensed by tools/summarize_profile.py), the short-K and mid-size tables of profiles/r5_policy/
set -u
mkdir -p gpurun_out/validate
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 2>&1 | tail -40 > gpurun_out/validate/gpu_tests_tail.txt
tail -4 gpurun_out/validate/gpu_tests_tail.txt
timeout 900 python bench.py > gpurun_out/validate/bench.json 2> gpurun_out/validate/bench.err
tail -c 400 gpurun_out/validate/bench.json
bash tools/profile.sh validate > gpurun_out/validate/profile.log 2>&1
timeout 600 python tools/ab.py --preset short_k --variants auto classic wide k64 OZIMMU_HIP_GEMM_KERNEL=k64,OZIMMU_HIP_K64_BREG=1 rocblas --legs 5 > gpurun_out/validate/short_k_policy_vs_forced.txt 2>&1
timeout 600 python tools/ab.py --preset mid --variants auto rocblas --legs 5 > gpurun_out/validate/mid_sizes_vs_rocblas.txt 2>&1
tail -30 gpurun_out/validate/short_k_policy_vs_forced.txt | cut -c1-260
