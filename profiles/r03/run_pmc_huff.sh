#!/bin/bash
# PMC passes over the device entropy kernels (batch of 256 x 1080p files, pixels left in device memory):
#   bash profiles/r03/run_pmc_huff.sh <tag> [444|420]
# counters in separate passes (8 SQ slots per pass); kernel-trace only beside them
set -u
TAG=${1:-r03}
SUB=${2:-444}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_huff_${TAG}_$SUB
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/e2e_bench.py --size 1920x1080 --sub $SUB --n 256 --threads 8 --modes device --no-pcie --repeat 1"
rocprofv3 -L > $O/counters_list.txt 2>&1 || true
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAVES_EQ_64 SQ_INSTS_VALU_MFMA_I8"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout -k 10 240 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/p$i -o p -- $CMD > $O/p$i.log 2>&1 || echo "pass $i failed"
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob("$O/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        out.setdefault(k, {}).update(v)
json.dump(out, open("$O/summary.json", "w"), indent=1)
for k, v in out.items():
    if "huff" in k: print(k, json.dumps(v))
PY
