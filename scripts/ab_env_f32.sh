#!/bin/bash
# Same-box A/B of environment settings of the fp32 configuration (B=64, three batches in flight AND one):
#   gpurun -- 'bash scripts/ab_env_f32.sh "" "POSERISK_CONV_PRIO=1" "POSERISK_CONV_PRIO=2"'
# each argument is one configuration (space-separated VAR=value pairs, "" = defaults); the list is run twice, in order.
# BATCH / LANES / EXTRA in the environment change the workload (defaults 64 / 3 / "").
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for round in 1 2; do
  for cfg in "$@"; do
    env $cfg timeout -k 10 200 python3 bench.py --batch ${BATCH:-64} --lanes ${LANES:-3} --cpu-frames 0 --steps 30 --repeats 3 $EXTRA > gpurun_out/ab_env.json
    python3 - "$cfg" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_env.json").read().strip().splitlines()[-1])
s = d["value_spread"]
r = d.get("roofline") or {}
print(f"[{sys.argv[1]:44s}] {d['value']:9.1f} frames/s ({s['min']:.0f} - {s['max']:.0f})  one batch {d.get('frames_per_s_one_batch_in_flight', 0):9.1f}"
      f"  conv {r.get('conv_ms_per_step', 0):.3f} ms  frac {r.get('frac', 0):.3f}  exec {r.get('mfma_executed_frac', 0):.3f}", flush=True)
PY
  done
done
