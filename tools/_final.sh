mkdir -p gpurun_out/r4u
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r4u/tests.txt 2>&1; tail -5 gpurun_out/r4u/tests.txt
timeout 900 python bench.py > gpurun_out/r4u/bench.json 2> gpurun_out/r4u/bench.err; cat gpurun_out/r4u/bench.json
bash tools/profile.sh r4u > gpurun_out/r4u/prof.log 2>&1; tail -5 gpurun_out/r4u/prof.log
