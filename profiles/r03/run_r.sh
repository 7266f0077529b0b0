#!/bin/bash
# round 3: one decode(bytes) end to end, default chunk size against 64-byte chunks
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r03r}
cd $R
timeout -k 10 400 python tools/single_latency.py > $O/${T}_single_128.txt 2>&1; echo "rc $?"
JPEGBLK_CHUNK_BYTES=64 timeout -k 10 400 python tools/single_latency.py > $O/${T}_single_64.txt 2>&1; echo "rc $?"
python - <<PY
import json
for tag in ("128", "64"):
    rows = [json.loads(l) for l in open("$O/${T}_single_%s.txt" % tag) if l.startswith('{"size"')]
    print(tag, [(r["size"], r["sub"], r["host_ms"], r["device_ms"]) for r in rows])
PY
