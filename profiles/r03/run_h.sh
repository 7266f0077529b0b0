#!/bin/bash
# round 3: where a batch of 128 1080p files spends its time (JPEGBLK_TIMING=3), arena output
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r03h}
cd $R
JPEGBLK_TIMING=3 timeout -k 10 400 python tools/e2e_bench.py --size 1920x1080 --sub 444 --n 1024 --threads 16 --modes arena --no-pcie --repeat 2 --stream 128 > $O/${T}_stream_arena.json 2> $O/${T}_stream_arena.err || { echo failed; tail -5 $O/${T}_stream_arena.err; exit 1; }
grep run_single $O/${T}_stream_arena.err | tail -60
