#!/bin/bash
# A/B of library builds inside one box:  bash profiles/r03/run_ab.sh <tag> "<lib> <lib> ..."   ("product" = libjpegblk.so)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-ab}
LIBS=${2:-product}
cd $R
for rep in 1 2; do
for lib in $LIBS; do
  if [ "$lib" = product ]; then unset JPEGBLK_LIB; else export JPEGBLK_LIB=$R/tools/ab/libjpegblk_h_$lib.so; fi
  for cb in 128 64; do
    export JPEGBLK_CHUNK_BYTES=$cb
    echo "== $lib chunk $cb rep $rep"
    timeout -k 10 200 python tools/single_latency.py --dri 0 --only 1920x1080 2>/dev/null | grep -v '"what"' | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(d['size'], d['sub'], 'device_ms', d['device_ms'])"
  done
  unset JPEGBLK_CHUNK_BYTES
  for sub in 444 420; do
    timeout -k 10 300 python tools/e2e_bench.py --size 1920x1080 --sub $sub --n 1024 --threads 16 --modes device --no-pcie --repeat 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    for r in d.get('decode_path', []): print('e2e $sub', r['output'][:6], r['images_per_s'], r['walls'])"
  done
done
done 2>&1 | tee $O/${T}_ab.txt
