#!/bin/bash
# round 3: small batches (128 x 1080p, 32 x 8192x8192) with device-resident and arena output, pass by pass
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r03q}
cd $R
for spec in "1920x1080 444 128" "8192x8192 420 32" "1920x1080 444 1024"; do
  set -- $spec
  timeout -k 10 400 python tools/e2e_bench.py --size $1 --sub $2 --n $3 --threads 16 --source writer --modes device,arena --no-pcie --repeat 6 > $O/${T}_$3.json 2> $O/${T}_$3.err || { echo "$spec failed"; tail -5 $O/${T}_$3.err; exit 1; }
  python - <<PY
import json
d = json.load(open("$O/${T}_$3.json"))
for r in d["decode_path"]:
    print("$spec", r["output"][:12], r["images_per_s"], r["walls"])
PY
done
