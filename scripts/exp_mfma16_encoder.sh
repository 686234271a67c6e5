#!/bin/bash
# Round 6: configs[2] (bf16 encoder, B=256, two batches in flight) on the shipped library against the build of the commit before
# the MFMA-shape conversion (scripts/ab_libs/old_shape.so), same box, A B A B.   gpurun -- 'bash scripts/exp_mfma16_encoder.sh'
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
ARGS="--precision bf16 --batch 256 --lanes 2 --steps 60 --warmup 10 --no-other-configs --cpu-frames 0 --repeats 3"
show() { python3 - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(f'value {d["value"]:.0f} f/s ({d["value_spread"]["min"]:.0f}..{d["value_spread"]["max"]:.0f})  ms/step {d["ms_per_step"]}  one batch in flight: conv {r["conv_ms_per_step"]} ms, frac {r["frac"]}  [{d["library"]["build"]}]')
PY
}
for i in 1 2; do
  echo "== old shape (32x32x16)"
  POSERISK_LIB_PATH=$PWD/scripts/ab_libs/old_shape.so timeout -k 10 200 python3 bench.py $ARGS > gpurun_out/r06_enc_old$i.json; show gpurun_out/r06_enc_old$i.json
  echo "== shipped"
  timeout -k 10 200 python3 bench.py $ARGS > gpurun_out/r06_enc_new$i.json; show gpurun_out/r06_enc_new$i.json
done
