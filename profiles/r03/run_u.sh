#!/bin/bash
# round 3: first device groups of a thread ramped (default) or full-sized (JPEGBLK_GROUP_RAMP=0), interleaved
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
for rep in 1 2; do
for ramp in 1 0; do
for spec in "1920x1080 444 128" "1920x1080 444 1024" "8192x8192 420 32"; do
  set -- $spec
  JPEGBLK_GROUP_RAMP=$ramp timeout -k 10 400 python tools/e2e_bench.py --size $1 --sub $2 --n $3 --threads 16 --source writer --modes device,arena --no-pcie --repeat 5 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for r in d['decode_path']: print('ramp $ramp', '$spec', r['output'][:12], r['images_per_s'], sorted(r['walls'])[:3])
"
done; done; done
