#!/bin/bash
# round 3: per-kernel totals of a host-output batch of 1,024 1080p files with the entropy stage on the device (the
# round-2 review quoted profiles/r02b/kernel_stats_e2e_device_entropy.csv: this is the same measurement on the final code)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03z_stats -o s -- python3 $R/tools/e2e_bench.py --size 1920x1080 --sub 444 --n 1024 --threads 16 --modes arena --no-pcie --repeat 3 > $O/r03z_stats.log 2>&1 || { echo failed; tail -3 $O/r03z_stats.log; exit 1; }
find $O/r03z_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r03z_kernel_stats_e2e.csv
head -12 $O/r03z_kernel_stats_e2e.csv | cut -c1-150
rm -rf $O/r03z_stats
