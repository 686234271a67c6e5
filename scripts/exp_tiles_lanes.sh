#!/bin/bash
# Larger conv tiles with SEVERAL batches in flight (round 5): single-lane layer tables said 128x64 / 64x128 lose at B=64
# (49 * 2^k tiles per layer cannot fill 256 CUs); with three batches in flight other lanes' kernels fill a launch's idle CUs,
# so the fetch saved per MFMA (0.75x at 128x64, 0.5x at 128x128) might pay there.  POSERISK_CONV_CFG forces a tile wherever it fits.
#   gpurun -- 'bash scripts/exp_tiles_lanes.sh > gpurun_out/r05_tiles_lanes.txt'
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
run() {
  echo "== $*"
  env "$@" python3 bench.py --no-other-configs --cpu-frames 0 --steps 100 --warmup 10 --repeats 3 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('value', d['value'], 'spread', d['value_spread']['min'], d['value_spread']['max'], 'one-lane', d.get('frames_per_s_one_batch_in_flight'), 'conv_ms', r['conv_ms_per_step'], 'frac', r['frac'])"
}
run X=0
run POSERISK_CONV_CFG=13
run POSERISK_CONV_CFG=12
run POSERISK_CONV_CFG=7
run POSERISK_CONV_CFG=10
run POSERISK_CONV_CFG=13 POSERISK_WINO_TILE=128x64
run X=0
