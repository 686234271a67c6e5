#!/bin/bash
# Same-box A/B of one environment switch of the bf16 encoder: gpurun -- 'bash scripts/ab_switch.sh POSERISK_BALANCED'
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
SW=${1:-POSERISK_BALANCED}
OUT=gpurun_out
for v in 1 0 1 0; do
  env $SW=$v timeout -k 10 200 python3 bench.py --precision bf16 --batch 256 --lanes 2 --cpu-frames 0 --no-roofline --steps 30 --repeats 3 > $OUT/ab_${SW}_$v.json
  python3 - <<PY
import json; d=json.loads(open("$OUT/ab_${SW}_$v.json").read().strip().splitlines()[-1]); print("$SW=$v", d["value"], d.get("value_spread"))
PY
done
