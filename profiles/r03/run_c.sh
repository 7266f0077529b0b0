#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r03c}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_huffman.py tests/test_gpu_batch.py -m gpu -x -q > $O/${T}_pytest.log 2>&1; echo "pytest rc $?"
tail -4 $O/${T}_pytest.log
timeout -k 10 300 python tools/single_latency.py --dri 0 > $O/${T}_single_latency.txt 2>&1; echo "single latency rc $?"
grep -v '"what"' $O/${T}_single_latency.txt | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(d['size'], d['sub'], 'host', d['host_ms'], 'device', d['device_ms'])"
for sub in 444 420; do
  timeout -k 10 300 python tools/e2e_bench.py --size 1920x1080 --sub $sub --n 1024 --threads 16 --modes arena,device --no-pcie --repeat 3 > $O/${T}_e2e_1080p_$sub.json 2> $O/${T}_e2e_1080p_$sub.err; echo "e2e $sub rc $?"
  python - <<PY
import json
for l in open("$O/${T}_e2e_1080p_$sub.json"):
    try: d=json.loads(l)
    except Exception: continue
    if isinstance(d,dict):
        for r in d.get("decode_path", []): print(r["output"], r["images_per_s"], r["walls"])
PY
done
cd /tmp && export TMPDIR=/tmp
for sub in 444 420; do
  rm -rf $O/${T}_trace_$sub
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/${T}_trace_$sub -o t -- python3 $R/tools/single_latency.py --only 1920x1080 --sub $sub --dri 0 > $O/${T}_trace_$sub.log 2>&1
  python3 $R/tools/trace_last_decode.py $O/${T}_trace_$sub > $O/${T}_launches_$sub.txt 2>&1
  cat $O/${T}_launches_$sub.txt
done
