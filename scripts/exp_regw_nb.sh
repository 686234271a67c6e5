#!/bin/bash
# conv1x1_regw_f32's unit shape (round 5): T 16-pixel tiles x NB 64-channel blocks per unit; every variant gives the same bits.
#   gpurun -- 'bash scripts/exp_regw_nb.sh > gpurun_out/r05_regw_nb.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
run() {
  echo "== $*"
  env "$@" python3 bench.py --no-other-configs --cpu-frames 0 --steps 100 --warmup 10 --repeats 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('value', d['value'], 'spread', d['value_spread']['min'], d['value_spread']['max'], 'one-lane', d.get('frames_per_s_one_batch_in_flight'), 'conv_ms', r['conv_ms_per_step'], 'frac', r['frac'], 'executed', r['mfma_executed_frac'])"
}
for rep in 1 2; do
run POSERISK_REGW_T=2 POSERISK_REGW_NB=1
run POSERISK_REGW_T=2 POSERISK_REGW_NB=2
run POSERISK_REGW_T=1 POSERISK_REGW_NB=2
run POSERISK_REGW_T=2 POSERISK_REGW_NB=4
run POSERISK_REGW_T=1 POSERISK_REGW_NB=4
done
for v in "2 1" "2 2" "1 2" "2 4" "1 4"; do
  set -- $v
  echo "== layer table T=$1 NB=$2"
  POSERISK_REGW_T=$1 POSERISK_REGW_NB=$2 python3 scripts/layer_table.py 2>/dev/null | grep -E "^L( 4| 7|11|14|15|17|18|20|21|23|27|28|30|31|33|34|36|37|39|40|42)|total"
done
