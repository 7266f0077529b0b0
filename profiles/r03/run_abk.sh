#!/bin/bash
# A/B of library builds by KERNEL time (rocprofv3 --kernel-trace --stats):  bash profiles/r03/run_abk.sh <tag> "<lib> ..." [chunk]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-abk}
LIBS=${2:-product}
CB=${3:-128}
cd /tmp && export TMPDIR=/tmp
export JPEGBLK_CHUNK_BYTES=$CB
for lib in $LIBS; do
  if [ "$lib" = product ]; then unset JPEGBLK_LIB; else export JPEGBLK_LIB=$R/tools/ab/libjpegblk_h_$lib.so; fi
  for what in "single 444" "single 420" "batch 444" "batch 420"; do
    set -- $what
    D=$O/${T}_${lib}_$1_$2
    rm -rf $D
    if [ $1 = single ]; then CMD="python3 $R/tools/single_latency.py --only 1920x1080 --sub $2 --dri 0"
    else CMD="python3 $R/tools/e2e_bench.py --size 1920x1080 --sub $2 --n 512 --threads 16 --modes device --no-pcie --repeat 2"; fi
    timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o s -- $CMD > $D.log 2>&1
    echo "== $lib $1 $2 (chunk $CB)"
    python3 - <<PY
import csv
for r in csv.DictReader(open("$D/s_kernel_stats.csv")):
    n = r["Name"].split("(")[0].replace("void ", "")
    if "huff" in n or "tile" in n: print(f"  {n:32s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
  done
done 2>&1 | tee $O/${T}_abk.txt
