#!/bin/bash
# round 3, first GPU call: new entropy kernels -- parity, single-image latency, launches of one decode
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_huffman.py -m gpu -x -q > $O/r03a_pytest_huff.log 2>&1; echo "pytest huff rc $?" | tee -a $O/r03a_pytest_huff.log
tail -5 $O/r03a_pytest_huff.log
timeout -k 10 300 python tools/single_latency.py > $O/r03a_single_latency.txt 2>&1; echo "single latency rc $?"
tail -22 $O/r03a_single_latency.txt
cd /tmp && export TMPDIR=/tmp
for sub in 444 420; do
  rm -rf $O/r03a_trace_$sub
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/r03a_trace_$sub -o t -- python3 $R/tools/single_latency.py --only 1920x1080 --sub $sub --dri 0 > $O/r03a_trace_$sub.log 2>&1
  python3 $R/tools/trace_last_decode.py $O/r03a_trace_$sub > $O/r03a_launches_$sub.txt 2>&1
  cat $O/r03a_launches_$sub.txt
done
