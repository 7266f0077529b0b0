#!/bin/bash
# round 3: what the device does during a device-resident batch of 1,024 1080p files (kernel + memory-copy trace, last pass)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r03m}
SRC=${2:-pil}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/${T}_trace -o t -- python3 $R/tools/e2e_bench.py --size 1920x1080 --sub 444 --n 1024 --threads 16 --source $SRC --modes device --no-pcie --repeat 3 > $O/${T}_dev.json 2> $O/${T}_dev.err || { echo failed; tail -5 $O/${T}_dev.err; exit 1; }
cd $R
python tools/timeline.py $O/${T}_trace --gap 20 > $O/${T}_timeline.txt 2>&1
cat $O/${T}_timeline.txt
python -c "
import json; d=json.load(open('$O/${T}_dev.json')); print([ (r['images_per_s'], r['walls']) for r in d['decode_path']])"
rm -rf $O/${T}_trace
