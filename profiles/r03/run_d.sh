#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r03h}
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/${T}_pytest.log 2>&1; echo "pytest rc $?"
tail -4 $O/${T}_pytest.log
e2e() {  # size sub n source
  python tools/e2e_bench.py --size $1 --sub $2 --n $3 --threads 16 --source $4 --modes arena --no-pcie --repeat 4 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    for r in d.get('decode_path', []): print('  ', r['output'][:6], r['images_per_s'], r['walls'])"
}
for rep in 1 2; do
for down in default 0; do
  if [ $down = default ]; then unset JPEGBLK_DEV_DOWN; else export JPEGBLK_DEV_DOWN=$down; fi
  echo "== downloads: $down (rep $rep)"
  echo " 1080p 444 x1024"; e2e 1920x1080 444 1024 pil
  echo " 1080p 444 x128"; e2e 1920x1080 444 128 pil
  echo " 8192 420 x64"; e2e 8192x8192 420 64 writer
done
done 2>&1 | tee $O/${T}_ab_downloads.txt
