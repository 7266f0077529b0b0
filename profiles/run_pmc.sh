#!/bin/bash
# Extra PMC passes for kernel diagnosis:  bash profiles/run_pmc.sh <tag>
set -u
TAG=${1:-pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-e2e"
rocprofv3 -L > $OUT/counters.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES_EQ_64 SQ_INSTS_FLAT SQ_INST_LEVEL_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o p$i -- $BENCH > $OUT/p$i.log 2>&1 || echo "pass $i failed: $set"
done
cd $R && python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "jb_tile_kernel" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: sum(v) / len(v) for k, v in acc.items()}
json.dump(res, open(out + "/pmc_avg.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
