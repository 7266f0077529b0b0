#!/bin/bash
# round 3: robustness of the rewritten entropy stage and the download thread -- mutation fuzzer, batch soak, pixel soak
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 200 python tools/huff_fuzz.py --seconds 120 --seed 37 > $O/r03_huff_fuzz_120s.txt 2>&1; echo "fuzz rc $?"; tail -2 $O/r03_huff_fuzz_120s.txt
timeout -k 10 330 python tools/batch_soak.py --seconds 240 --seed 29 > $O/r03_batch_soak_240s.txt 2>&1; echo "soak rc $?"; tail -3 $O/r03_batch_soak_240s.txt
timeout -k 10 160 python tools/stress.py --seconds 120 --seed 13 > $O/r03_stress_120s.txt 2>&1; echo "stress rc $?"; tail -2 $O/r03_stress_120s.txt
