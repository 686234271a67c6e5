#!/bin/bash
# third pass: (T, NB) = (1, 1) -- half-size units, finer shares -- for the plain layers and the Winograd GEMMs
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for v in "1 2 2 1" "1 1 2 1" "1 2 1 1"; do
  set -- $v
  echo "== layer table plain T=$1 NB=$2 wino T=$3 NB=$4"
  POSERISK_REGW_T=$1 POSERISK_REGW_NB=$2 POSERISK_REGW_WT=$3 POSERISK_REGW_WNB=$4 python3 scripts/layer_table.py 2>/dev/null | grep -E "^L( 4| 7|11|15|16|17|19|20|22|23|29|30|32|33|35|36|38|39|41|42)|total"
done
run() {
  echo "== $*"
  env "$@" python3 bench.py --no-other-configs --cpu-frames 0 --steps 100 --warmup 10 --repeats 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('value', d['value'], 'spread', d['value_spread']['min'], d['value_spread']['max'], 'one-lane', d.get('frames_per_s_one_batch_in_flight'), 'conv_ms', r['conv_ms_per_step'], 'frac', r['frac'])"
}
for rep in 1 2; do
run X=0
run POSERISK_REGW_WT=1 POSERISK_REGW_WNB=1
run POSERISK_REGW_T=1 POSERISK_REGW_NB=1
done
