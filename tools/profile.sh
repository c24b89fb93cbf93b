#!/bin/bash
# tools/profile.sh <tag> [bench args...] — run on the GPU box (via gpurun) from the repo root.
# Collects (1) rocprofv3 --kernel-trace --stats of bench.py, (2) PMC passes, each in its own run
# (never combined with trace domains other than --kernel-trace), into gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r1}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extra --no-traffic --no-ceiling $*"

timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log

pmc() { # name, counters...
  local name=$1; shift
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o pmc -- $BENCH > $OUT/pmc_$name.log 2>&1 \
    || echo "pmc pass $name failed (see $OUT/pmc_$name.log)"
}
pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 GRBM_GUI_ACTIVE
pmc sq2 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU
pmc tcc1 TCC_HIT_sum TCC_MISS_sum
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
ls -R $OUT | head -60
