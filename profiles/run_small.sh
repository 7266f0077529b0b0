#!/bin/bash
# Small / single-image regime (BASELINE configs 2, 3 and the reference's bundled-image size):
#   bash profiles/run_small.sh <tag>
# per workload: rocprofv3 kernel trace, then separate --pmc passes for FETCH_SIZE, WRITE_SIZE and the
# L2<->fabric request mix (32-B vs 64-B reads and writes: partial-line writes show up there).
# Single images run COLD: --sets rotates over > 512 MiB of distinct buffers.
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-configs --steps 200 --warmup 20 --precondition 100"
run_one() {  # name, bench args, algorithmic bytes per launch
  local NAME=$1 ARGS=$2 ALG=$3
  local OUT=$R/gpurun_out/prof_${TAG}_$NAME
  mkdir -p $OUT
  local BENCH="python3 $R/bench.py --no-e2e $COMMON $ARGS"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1 || echo "trace failed: $NAME"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || echo "pmc fetch failed: $NAME"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $BENCH > $OUT/pmc_write.log 2>&1 || echo "pmc write failed: $NAME"
  rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/pmc_req -o req -- $BENCH > $OUT/pmc_req.log 2>&1 || echo "pmc req failed: $NAME"
  (cd $R && JB_ALG_BYTES=$ALG python3 profiles/summarize_profile.py $OUT ${TAG}_$NAME 200 > $OUT/summary.log 2>&1; tail -25 $OUT/summary.log)
}
run_one 679x451_420_x512 "--workload 679x451-420 --images-per-step 512" $((512 * (7482 * 128 + 679 * 451 * 3)))
run_one 1080p_444_x1_cold "--workload 1920x1080-444 --images-per-step 1 --sets 32" $((97200 * 128 + 1920 * 1080 * 3))
run_one 4096_420_x1_cold "--workload 4096x4096-420 --images-per-step 1 --sets 8" $((393216 * 128 + 4096 * 4096 * 3))
