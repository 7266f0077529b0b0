#!/bin/bash
# round 3: one download engine per device -- batch tests, then batches of 128 one after the other / two in flight
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r03i2}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_huffman.py -x -q > $O/${T}_pytest.log 2>&1; rc=$?; echo "pytest rc $rc"
tail -3 $O/${T}_pytest.log
[ $rc -eq 0 ] || exit 1
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
timeout -k 10 300 python tools/two_decoders.py > $O/${T}_two_decoders.txt 2>&1 || { echo "two_decoders failed"; tail -3 $O/${T}_two_decoders.txt; }
python - <<PY
import json
d = json.load(open("$O/${T}_8192.json"))
for r in d["decode_path"]:
    print("8192", r["output"], r["images_per_s"], r["walls"], r.get("stream"))
PY
tail -12 $O/${T}_two_decoders.txt
