#!/bin/bash
# round 3: batches of 128 / 8 files one after the other against two in flight, after a change to the download order
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r03k}
cd $R
for mode in arena malloc; do
  timeout -k 10 400 python tools/e2e_bench.py --size 1920x1080 --sub 444 --n 1024 --threads 16 --modes $mode --no-pcie --repeat 5 --stream 128 > $O/${T}_stream_$mode.json 2> $O/${T}_stream_$mode.err || { echo "stream $mode failed"; tail -5 $O/${T}_stream_$mode.err; exit 1; }
  python - <<PY
import json
d = json.load(open("$O/${T}_stream_$mode.json"))
for r in d["decode_path"]:
    print("$mode", r["threads"], "whole batch", r["images_per_s"], r["walls"], "stream", r.get("stream"))
PY
done
timeout -k 10 400 python tools/e2e_bench.py --size 8192x8192 --sub 420 --n 64 --threads 16 --source writer --modes arena --no-pcie --repeat 3 --stream 8 > $O/${T}_8192.json 2> $O/${T}_8192.err || exit 1
python - <<PY
import json
d = json.load(open("$O/${T}_8192.json"))
for r in d["decode_path"]:
    print("8192", r["output"], r["images_per_s"], r["walls"], r.get("stream"))
PY
