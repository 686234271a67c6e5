#!/bin/bash
# persistent workgroups per CU of conv1x1_regw_f32 against batches in flight, with round 5's smaller units (16 / 32 KB of LDS per workgroup)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
run() {
  echo "== $*"
  L=$1; shift
  env "$@" python3 bench.py --lanes $L --no-other-configs --no-roofline --cpu-frames 0 --steps 100 --warmup 10 --repeats 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'spread', d['value_spread']['min'], d['value_spread']['max'], 'one-lane', d.get('frames_per_s_one_batch_in_flight'))"
}
for rep in 1 2; do
run 3 X=0
run 3 POSERISK_REGW_PER_CU=2
run 3 POSERISK_REGW_PER_CU=3
run 2 X=0
run 4 X=0
run 4 POSERISK_REGW_PER_CU=2
done
