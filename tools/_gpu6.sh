set -u
mkdir -p gpurun_out/r5f
export TMPDIR=/tmp
timeout 900 python tools/policy_fit.py collect gpurun_out/r5f/policy_q2_shortk_seed5.jsonl --kind shortk --count 80 --seed 5 --modes 4 6 8 9 10 11 12 > gpurun_out/r5f/c1.log 2>&1
timeout 900 python tools/policy_fit.py collect gpurun_out/r5f/policy_q2_seed6.jsonl --count 90 --seed 6 --modes 4 6 8 9 10 11 12 > gpurun_out/r5f/c2.log 2>&1
timeout 600 python tools/policy_fit.py collect gpurun_out/r5f/policy_q2_shortk_seed7.jsonl --kind shortk --count 40 --seed 7 --modes 4 6 8 9 10 11 12 > gpurun_out/r5f/c3.log 2>&1
timeout 900 python tools/policy_fit.py collect gpurun_out/r5f/policy_q2_seed8.jsonl --count 90 --seed 8 --modes 4 6 8 9 10 11 12 > gpurun_out/r5f/c4.log 2>&1
timeout 900 python tools/policy_fit.py collect gpurun_out/r5f/policy_q2_seed9.jsonl --count 120 --seed 9 --modes 4 6 8 9 10 11 12 > gpurun_out/r5f/c5.log 2>&1
wc -l gpurun_out/r5f/*.jsonl
