# tools/validate_gpu.sh - one GPU-box call (gpurun -- bash tools/validate_gpu.sh): the whole GPU suite, bench.py, rocprofv3 kernel stats + PMC
# (tools/profile.sh -> gpurun_out/prof_validate, condensed by tools/summarize_profile.py), the short-K and mid-size tables of profiles/r5_policy/, the blocked LU through the preload
set -u
mkdir -p gpurun_out/validate
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 2>&1 | tail -40 > gpurun_out/validate/gpu_tests_tail.txt
tail -4 gpurun_out/validate/gpu_tests_tail.txt
# one leg of the parity tests under the PRODUCTION kernel choice (the suite above runs with OZIMMU_HIP_AUTOTUNE=0, bench.py
# with the default): measured choice on - the session-wide handle meets the same shapes again and again, so explorations and decisions
# happen all through it.  Excluded: the tests that assert WHICH kernel the cost model picks (a timing may decide otherwise by design) and
# the tuner's own tests; the switches are still followed per call (the tests force split forms and kernels through them).
OZIMMU_HIP_AUTOTUNE=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_gpu_zgemm.py tests/test_gpu_k2_kernel.py \
  tests/test_gpu_wide_kernel.py tests/test_gpu_forced_kernels.py tests/test_gpu_one_launch.py tests/test_gpu_robustness.py -m gpu -q --maxfail=30 \
  -k "not policy and not picks and not default_where" 2>&1 | tail -30 > gpurun_out/validate/gpu_tests_tuner_on_tail.txt
tail -3 gpurun_out/validate/gpu_tests_tuner_on_tail.txt
timeout 900 python bench.py > gpurun_out/validate/bench.json 2> gpurun_out/validate/bench.err
tail -c 400 gpurun_out/validate/bench.json
bash tools/profile.sh validate > gpurun_out/validate/profile.log 2>&1
timeout 600 python tools/ab.py --preset short_k --variants auto classic wide k64 OZIMMU_HIP_GEMM_KERNEL=k64,OZIMMU_HIP_K64_BREG=1 rocblas --legs 5 > gpurun_out/validate/short_k_policy_vs_forced.txt 2>&1
timeout 600 python tools/ab.py --preset mid --variants auto rocblas --legs 5 > gpurun_out/validate/mid_sizes_vs_rocblas.txt 2>&1
tail -30 gpurun_out/validate/short_k_policy_vs_forced.txt | cut -c1-260
timeout 900 python tools/lu_preload.py --n 16384 --nb 512 > gpurun_out/validate/lu_16384.txt 2>&1
timeout 1200 python tools/lu_preload.py --n 32768 --nb 512 > gpurun_out/validate/lu_32768.txt 2>&1
cat gpurun_out/validate/lu_16384.txt gpurun_out/validate/lu_32768.txt
