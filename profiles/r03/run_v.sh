#!/bin/bash
# round 3: size of a device-entropy group (JPEGBLK_DEV_GROUP_MB), interleaved on one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
for mb in 96 48 192 384; do
for spec in "1920x1080 444 128" "1920x1080 444 1024"; do
  set -- $spec
  JPEGBLK_DEV_GROUP_MB=$mb timeout -k 10 400 python tools/e2e_bench.py --size $1 --sub $2 --n $3 --threads 16 --source writer --modes device,arena --no-pcie --repeat 5 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for r in d['decode_path']: print('group_mb $mb', '$spec', r['output'][:12], r['images_per_s'], sorted(r['walls'])[:3])
"
done; done; done
