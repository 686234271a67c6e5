#!/bin/bash
# Same-box A/B of one environment switch on the fp32 headline (B = 64, three batches in flight) and on the per-layer table
# (one batch in flight).   gpurun -- 'bash scripts/exp_env_ab.sh POSERISK_WINO_VEC=4'
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
SW="$1"; shift || true
ARGS="--steps 100 --warmup 10 --no-other-configs --cpu-frames 0 --repeats 3 $*"
show() { python3 - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print(f'value {d["value"]:.0f} f/s ({d["value_spread"]["min"]:.0f}..{d["value_spread"]["max"]:.0f})  one batch in flight: {d.get("frames_per_s_one_batch_in_flight")} f/s, conv {r.get("conv_ms_per_step")} ms, frac {r.get("frac")}')
PY
}
for i in 1 2; do
  echo "== default"; timeout -k 10 200 python3 bench.py $ARGS > gpurun_out/ab_A$i.json; show gpurun_out/ab_A$i.json
  echo "== $SW"; env $SW timeout -k 10 200 python3 bench.py $ARGS > gpurun_out/ab_B$i.json; show gpurun_out/ab_B$i.json
done
