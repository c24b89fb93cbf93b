set -u
mkdir -p gpurun_out/r5c
export TMPDIR=/tmp
timeout 900 python tools/policy_fit.py collect gpurun_out/r5c/policy_q_shortk_seed5.jsonl --kind shortk --count 80 --seed 5 --modes 4 6 8 9 10 11 12 > gpurun_out/r5c/collect_shortk.log 2>&1
timeout 900 python tools/policy_fit.py collect gpurun_out/r5c/policy_q_seed6.jsonl --count 90 --seed 6 --modes 4 6 8 9 10 11 12 > gpurun_out/r5c/collect_mixed.log 2>&1
timeout 600 python tools/policy_fit.py collect gpurun_out/r5c/policy_q_shortk_seed7.jsonl --kind shortk --count 40 --seed 7 --modes 4 6 8 9 10 11 12 > gpurun_out/r5c/collect_shortk_heldout.log 2>&1
wc -l gpurun_out/r5c/*.jsonl
