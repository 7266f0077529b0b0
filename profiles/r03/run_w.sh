#!/bin/bash
# round 3: chunk size of the device entropy decoder and the host-thread count in batches, interleaved on one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
for cb in 128 64; do
for spec in "1920x1080 444 1024 writer" "1920x1080 420 1024 writer" "1920x1080 444 1024 pil"; do
  set -- $spec
  JPEGBLK_CHUNK_BYTES=$cb timeout -k 10 400 python tools/e2e_bench.py --size $1 --sub $2 --n $3 --threads 16 --source $4 --modes device --no-pcie --repeat 5 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for r in d['decode_path']: print('chunk $cb', '$spec', r['output'][:12], r['images_per_s'], sorted(r['walls'])[:3])
"
done; done
for th in 8 16 24 32; do
  timeout -k 10 400 python tools/e2e_bench.py --size 1920x1080 --sub 444 --n 1024 --threads $th --source writer --modes device,arena --no-pcie --repeat 5 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for r in d['decode_path']: print('threads $th', r['output'][:12], r['images_per_s'], sorted(r['walls'])[:3])
"
done; done
