#!/bin/bash
# Profile recipe (run on the GPU box through gpurun):  bash profiles/run_profile.sh <tag>
# 1. rocprofv3 --kernel-trace --stats of the default bench command  -> per-kernel durations
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE cannot share a pass on gfx950)
# Outputs land in gpurun_out/prof_<tag>/ ; profiles/summarize_profile.py condenses them.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the default bench command (400 untimed pre-conditioning launches, 20 warm-up, 200 timed steps)
BENCH="python3 $R/bench.py --no-cpu-baseline --no-configs --no-e2e ${JB_BENCH_ARGS:-}"   # JB_BENCH_ARGS="--workload 4096x4096-420" profiles another config
TIMED=200
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1 || echo "trace failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || echo "pmc fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $BENCH > $OUT/pmc_write.log 2>&1 || echo "pmc write failed"
cd $R && python3 profiles/summarize_profile.py $OUT $TAG $TIMED
