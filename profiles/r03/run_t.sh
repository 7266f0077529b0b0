#!/bin/bash
# round 3: what the link does with the entropy stage on the HOST threads (BASELINE config 5 as worded): 64 x 8192x8192 4:2:0, 64 threads
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r03t}
TH=${2:-64}
cd /tmp && export TMPDIR=/tmp
JPEGBLK_GPU_HUFFMAN=0 timeout -k 10 500 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/${T}_trace -o t -- python3 $R/tools/e2e_bench.py --size 8192x8192 --sub 420 --n 64 --threads $TH --source writer --modes arena --no-pcie --repeat 2 > $O/${T}_host.json 2> $O/${T}_host.err || { echo failed; tail -5 $O/${T}_host.err; exit 1; }
cd $R
python tools/timeline.py $O/${T}_trace --gap 30 > $O/${T}_timeline.txt 2>&1
cat $O/${T}_timeline.txt
python -c "
import json; d=json.load(open('$O/${T}_host.json')); print([(r['threads'], r['images_per_s'], r['walls'], r['entropy_cpu_s'], r['submit_wait_s']) for r in d['decode_path']])"
rm -rf $O/${T}_trace
