#!/bin/bash
# Timing ablations of the fused kernel (experiment builds under tools/ab/, results are WRONG by
# design: a stage is skipped at run time through a condition the compiler cannot fold).
#   here (no GPU needed):  bash tools/ablate.sh build
#   on the GPU box:        bash tools/ablate.sh            [JB_BENCH_ARGS="--workload 4096x4096-420"]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
if [ "${1:-}" = build ]; then
  bash $R/tools/build_variant.sh NO_LOAD -DJB_EXP_NO_LOAD
  bash $R/tools/build_variant.sh NO_IDCT -DJB_EXP_NO_IDCT
  bash $R/tools/build_variant.sh NO_COLOUR -DJB_EXP_NO_COLOUR
  bash $R/tools/build_variant.sh NO_MATH -DJB_EXP_NO_IDCT -DJB_EXP_NO_COLOUR
  bash $R/tools/build_variant.sh NO_STORE -DJB_EXP_NO_STORE
  exit 0
fi
for v in "" NO_LOAD NO_IDCT NO_COLOUR NO_MATH NO_STORE; do
  lib=$R/jpeg_decoder_amd/libjpegblk.so; [ -n "$v" ] && lib=$R/tools/ab/libjpegblk_$v.so
  JPEGBLK_LIB=$lib timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --no-e2e ${JB_BENCH_ARGS:-} 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)['roofline']; print('%-10s mean %.1f us  min %.1f us  (%.0f GB/s-equivalent)' % ('${v:-full}', d['kernel_ms_mean']*1e3, d['kernel_ms_min']*1e3, d['achieved']))
"
done
