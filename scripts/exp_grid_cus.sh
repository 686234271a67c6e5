#!/bin/bash
# persistent kernels sized for a share of the chip, several batches in flight (bf16 B=256; fp32 B=64)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
run() {
  echo "== $*"
  A="$1"; shift
  env "$@" python3 bench.py $A --no-other-configs --no-roofline --cpu-frames 0 --steps 100 --warmup 10 --repeats 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'spread', d['value_spread']['min'], d['value_spread']['max'], 'one-lane', d.get('frames_per_s_one_batch_in_flight'))"
}
for rep in 1 2; do
run "--precision bf16 --batch 256 --lanes 2" X=0
run "--precision bf16 --batch 256 --lanes 2" POSERISK_GRID_CUS=128
run "--precision bf16 --batch 256 --lanes 2" POSERISK_GRID_CUS=192
run "--precision bf16 --batch 256 --lanes 3" POSERISK_GRID_CUS=128
run "--precision bf16 --batch 256 --lanes 3" POSERISK_GRID_CUS=96
run "--precision bf16 --batch 256 --lanes 4" POSERISK_GRID_CUS=64
done
run "--lanes 3" POSERISK_GRID_CUS=128
run "--lanes 3" POSERISK_GRID_CUS=192
run "--lanes 3" X=0
