#!/bin/bash
# A/B of library builds on ONE GPU box, interleaved so that box-to-box and power-state drift
# cancel:  bash tools/ab_bench.sh "<lib A> <lib B> ..." "<workload[:images]> ..." [reps]
# Each library is a path to a libjpegblk.so build (JPEGBLK_LIB picks it up, api.py).
set -u
LIBS=${1:?libs}
WLS=${2:-"4096x4096-444 1920x1080-444:128 4096x4096-420"}
REPS=${3:-3}
for rep in $(seq $REPS); do
  for w in $WLS; do
    # workload[:images per launch[:rotating buffer sets]]
    IFS=: read -r wl n sets <<< "$w"; n=${n:-8}; sets=${sets:-1}
    for lib in $LIBS; do
      JPEGBLK_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-configs --no-e2e --workload $wl --images-per-step $n --sets $sets ${AB_BENCH_ARGS:-} 2>/dev/null |
        python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('%-28s %-26s kernel %.2f us (median %.2f, min %.2f)  %.0f GB/s  frac %.3f' % ('$(basename $lib)', '$wl x$n sets $sets', r['kernel_ms_mean'] * 1e3, r['kernel_ms_median'] * 1e3, r['kernel_ms_min'] * 1e3, r['achieved'], r['frac']))
"
    done
  done
done
