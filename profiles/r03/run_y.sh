#!/bin/bash
# round 3: hardware queues asked of the HIP runtime (JPEGBLK_HW_QUEUES, read when the library loads), interleaved on one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
for q in 16 4 8 32; do
  JPEGBLK_HW_QUEUES=$q timeout -k 10 400 python tools/e2e_bench.py --size 1920x1080 --sub 444 --n 1024 --threads 16 --source writer --modes device,arena --no-pcie --repeat 5 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for r in d['decode_path']: print('hw_queues $q', r['output'][:12], r['images_per_s'], sorted(r['walls'])[:3])
"
done; done
