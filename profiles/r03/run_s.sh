#!/bin/bash
# round 3: timeline of batch runs (JPEGBLK_TIMING=3) on the final binaries: 1,024 and 128 files, arena output
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
JPEGBLK_TIMING=3 timeout -k 10 400 python tools/e2e_bench.py --size 1920x1080 --sub 444 --n 1024 --threads 16 --modes arena --no-pcie --repeat 3 --stream 128 > $O/r03s.json 2> $O/r03s.err || { echo failed; tail -5 $O/r03s.err; exit 1; }
grep "run_single 1024" $O/r03s.err | tail -3
grep "run_single 128" $O/r03s.err | tail -12
