#!/bin/bash
# round 3: the staged store stage (a -DJB_LAB build: bash tools/build_variant.sh lab) -- the product's store-path parity test, then
# the staged stage's bytes and launch times against the product path's
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r03n}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "unaligned_output" > $O/${T}_pytest.log 2>&1; rc=$?; echo "pytest rc $rc"
tail -15 $O/${T}_pytest.log
[ $rc -eq 0 ] || exit 1
JPEGBLK_LIB=$R/tools/ab/libjpegblk_lab.so timeout -k 10 600 python profiles/r03/probe_staged.py > $O/${T}_probe.txt 2>&1; echo "probe rc $?"
tail -12 $O/${T}_probe.txt
