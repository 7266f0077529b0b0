#!/bin/bash
# round 3: batch throughput with the new entropy kernels + kernel stats
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_huffman.py tests/test_gpu_batch.py -m gpu -x -q > $O/r03b_pytest.log 2>&1; echo "pytest rc $?"
tail -4 $O/r03b_pytest.log
for sub in 444 420; do
  timeout -k 10 300 python tools/e2e_bench.py --size 1920x1080 --sub $sub --n 1024 --threads 16 --modes arena,device --no-pcie --repeat 3 > $O/r03b_e2e_1080p_$sub.json 2> $O/r03b_e2e_1080p_$sub.err; echo "e2e $sub rc $?"
  python - <<PY
import json
for l in open("$O/r03b_e2e_1080p_$sub.json"):
    try: d=json.loads(l)
    except Exception: continue
    if isinstance(d,dict):
        for k,v in d.items():
            if isinstance(v,list):
                for r in v:
                    if isinstance(r,dict): print({kk:r[kk] for kk in r if kk in ("mode","threads","images_per_s","gpixel_per_s","walls_s","entropy_on_device","pixels_checked")})
PY
done
timeout -k 10 300 python tools/e2e_bench.py --size 8192x8192 --sub 420 --n 64 --threads 16 --source writer --modes arena,device --no-pcie --repeat 3 > $O/r03b_e2e_8192.json 2> $O/r03b_e2e_8192.err; echo "e2e 8192 rc $?"
tail -c 1500 $O/r03b_e2e_8192.json
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r03b_stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03b_stats -o s -- python3 $R/tools/e2e_bench.py --size 1920x1080 --sub 444 --n 1024 --threads 16 --modes device --no-pcie --repeat 2 > $O/r03b_stats.log 2>&1
head -12 $O/r03b_stats/s_kernel_stats.csv
