#!/bin/bash
# fc_rows16_f32's output tiles per workgroup (round 5; same bits): bf16 B=256 (two batches in flight and one), fp32 B=256, fp32 B=64
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
for S in 11 22 42 41 21 12; do
run "--precision bf16 --batch 256 --lanes 2" POSERISK_FC_SHAPE=$S
done
done
for S in 11 22 42; do
run "--batch 256 --lanes 2" POSERISK_FC_SHAPE=$S
done
for S in 11 22 21 12; do
run "--lanes 3" POSERISK_FC_SHAPE=$S
done
