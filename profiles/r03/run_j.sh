#!/bin/bash
# round 3: what the link does while batches of 128 stream two in flight (kernel + memory-copy trace, last pass)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r03j}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/${T}_trace -o t -- python3 $R/tools/e2e_bench.py --size 1920x1080 --sub 444 --n 1024 --threads 16 --modes arena --no-pcie --repeat 2 --stream 128 > $O/${T}_stream.json 2> $O/${T}_stream.err || { echo failed; tail -5 $O/${T}_stream.err; exit 1; }
cd $R
python tools/timeline.py $O/${T}_trace --gap 30 > $O/${T}_timeline.txt 2>&1
cat $O/${T}_timeline.txt
rm -rf $O/${T}_trace
