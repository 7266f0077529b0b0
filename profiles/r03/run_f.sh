#!/bin/bash
# round 3: submit / collect (two batches in flight) -- its tests, then batches of 128 1080p files one after the other
# against streamed; and the device-resident rate on writer-made files after the de-stuffing change
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r03f}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py -x -q -k "submit_collect or device_resident" > $O/${T}_pytest.log 2>&1; rc=$?; echo "pytest rc $rc"
tail -5 $O/${T}_pytest.log
[ $rc -eq 0 ] || exit 1
for mode in arena malloc; do
  timeout -k 10 400 python tools/e2e_bench.py --size 1920x1080 --sub 444 --n 1024 --threads 16 --modes $mode --no-pcie --repeat 3 --stream 128 > $O/${T}_stream_$mode.json 2> $O/${T}_stream_$mode.err || { echo "stream $mode failed"; tail -5 $O/${T}_stream_$mode.err; exit 1; }
  python - <<PY
import json
d = json.load(open("$O/${T}_stream_$mode.json"))
for r in d["decode_path"]:
    print("$mode", r["threads"], "whole batch", r["images_per_s"], r["walls"], "stream", r.get("stream"))
PY
done
for src in writer pil; do
  timeout -k 10 400 python tools/e2e_bench.py --size 1920x1080 --sub 444 --n 1024 --threads 16 --source $src --modes device --no-pcie --repeat 4 > $O/${T}_device_$src.json 2> $O/${T}_device_$src.err || { echo "device $src failed"; tail -5 $O/${T}_device_$src.err; exit 1; }
  python - <<PY
import json
d = json.load(open("$O/${T}_device_$src.json"))
for r in d["decode_path"]:
    print("$src", d["file_kbytes_mean"], "KB/file", r["output"], r["images_per_s"], r["walls"], "entropy_cpu_s", r["entropy_cpu_s"])
PY
done
