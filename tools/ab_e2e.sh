#!/bin/bash
# Interleaved A/B of environment-variable variants of the batch decoder on one GPU box:
#   tools/ab_e2e.sh "<VAR=a VAR=b ...>" <rounds> -- <e2e_bench.py arguments>
# every variant runs <rounds> times, interleaved; prints images/s per run (best of --repeat passes).
variants="$1"; rounds="$2"; shift 3
for r in $(seq 1 "$rounds"); do
  for v in $variants; do
    out=$(env ${v//,/ } timeout -k 10 200 python tools/e2e_bench.py "$@" --no-pcie 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin)['decode_path'][0]; print(d['images_per_s'], d['walls'], 'dev' if d['entropy_on_device'] else 'host')")
    echo "round $r  $v  $out"
  done
done
