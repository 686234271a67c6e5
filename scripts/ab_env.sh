#!/bin/bash
# Same-box A/B of environment settings of the bf16 encoder (B=256, two batches in flight):
#   gpurun -- 'bash scripts/ab_env.sh "POSERISK_FUSE_BOTTLENECK2=0" "POSERISK_B128_LEAD=0" "POSERISK_B128_LEAD=2"'
# each argument is one configuration (space-separated VAR=value pairs, "" = defaults); the list is run twice, in order.
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
for round in 1 2; do
  for cfg in "$@"; do
    env $cfg timeout -k 10 200 python3 bench.py --precision bf16 --batch 256 --lanes 2 --cpu-frames 0 --no-roofline --steps 30 --repeats 3 > gpurun_out/ab_env.json
    python3 - "$cfg" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_env.json").read().strip().splitlines()[-1])
s = d["value_spread"]
print(f"[{sys.argv[1]:40s}] {d['value']:9.1f} frames/s  ({s['min']:.0f} - {s['max']:.0f})")
PY
  done
done
