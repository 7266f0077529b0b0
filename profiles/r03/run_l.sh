#!/bin/bash
# round 3: the small-grid 4:4:4 kernel -- parity, then its launch time against the 192-lane kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r03l}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "small_grid" > $O/${T}_pytest.log 2>&1; rc=$?; echo "pytest rc $rc"
tail -15 $O/${T}_pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python profiles/r03/probe_small_grid.py > $O/${T}_probe.txt 2>&1; echo "probe rc $?"
cat $O/${T}_probe.txt | tail -12
