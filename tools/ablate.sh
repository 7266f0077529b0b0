#!/bin/bash
# Timing ablations of the fused kernel (experiment builds in tools/, results are WRONG by design).
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "" NO_LOAD NO_IDCT NO_COLOUR NO_MATH NO_STORE; do
  lib=$R/jpeg_decoder_amd/libjpegblk.so; [ -n "$v" ] && lib=$R/tools/libjpegblk_$v.so
  JPEGBLK_LIB=$lib timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline $JB_BENCH_ARGS 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)['roofline']; print('%-10s mean %.1f us  min %.1f us  (%.0f GB/s-equivalent)' % ('${v:-full}', d['kernel_ms_mean']*1e3, d['kernel_ms_min']*1e3, d['achieved']))
"
done
