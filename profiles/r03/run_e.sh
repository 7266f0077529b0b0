#!/bin/bash
# round 3: full GPU suite, the headline profile (kernel trace + FETCH/WRITE PMC), PMC of the entropy kernels, bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r03i}
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/${T}_pytest.log 2>&1; echo "pytest rc $?"
tail -4 $O/${T}_pytest.log
bash profiles/run_profile.sh r03 > $O/${T}_profile.log 2>&1; echo "profile rc $?"; tail -3 $O/${T}_profile.log
bash profiles/r03/run_pmc_huff.sh ${T} 444 > $O/${T}_pmc444.log 2>&1; echo "pmc 444 rc $?"
bash profiles/r03/run_pmc_huff.sh ${T} 420 > $O/${T}_pmc420.log 2>&1; echo "pmc 420 rc $?"
timeout -k 10 900 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads([l for l in open("$O/${T}_bench.json") if l.startswith("{")][0])
print({k: d[k] for k in ("metric", "value", "ms_per_step", "roofline")})
for k, v in d["end_to_end"]["runs"].items():
    for form in ("weak", "strong"):
        for m, r in v[form].items():
            if isinstance(r, dict): print(k, form, m, r["images_per_s"], r.get("images_per_s_median"), r["walls"])
print({k: (v.get("kernel_us"), v.get("frac")) for k, v in d.get("configs", {}).items()})
PY
