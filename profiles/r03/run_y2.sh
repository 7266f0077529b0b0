#!/bin/bash
# round 3: hardware queues (JPEGBLK_HW_QUEUES) with the entropy stage on the HOST threads, and single-image submissions
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
for q in 16 4; do
  JPEGBLK_GPU_HUFFMAN=0 JPEGBLK_HW_QUEUES=$q timeout -k 10 400 python tools/e2e_bench.py --size 1920x1080 --sub 444 --n 1024 --threads 16,64 --source writer --modes arena,malloc --no-pcie --repeat 3 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for r in d['decode_path']: print('host-entropy hw_queues $q', r['output'][:12], r['threads'], r['images_per_s'], sorted(r['walls'])[:3])
"
  JPEGBLK_GPU_HUFFMAN=0 JPEGBLK_GROUP_MB=0 JPEGBLK_HW_QUEUES=$q timeout -k 10 400 python tools/e2e_bench.py --size 679x451 --sub 420 --n 4096 --threads 16 --source writer --modes arena --no-pcie --repeat 3 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for r in d['decode_path']: print('host-entropy one image per submission hw_queues $q 679x451', r['threads'], r['images_per_s'], sorted(r['walls'])[:3])
"
  JPEGBLK_HW_QUEUES=$q timeout -k 10 400 python tools/e2e_bench.py --size 679x451 --sub 420 --n 4096 --threads 16 --source writer --modes arena,device --no-pcie --repeat 3 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for r in d['decode_path']: print('device-entropy hw_queues $q 679x451', r['output'][:12], r['images_per_s'], sorted(r['walls'])[:3])
"
done; done
