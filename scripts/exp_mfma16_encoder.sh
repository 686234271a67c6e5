#!/bin/bash
# Round 6, step 1b: the whole bf16 encoder (configs[2]) with every 32x32x16 MFMA step issued as two 16x16x32 ones
# (PR_EXPERIMENT=16, results wrong) against the shipped library, same box, A B A B.
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
ARGS="--precision bf16 --batch 256 --lanes 2 --steps 60 --warmup 10 --no-other-configs --cpu-frames 0 --repeats 3"
for i in 1 2; do
  echo "== shipped (32x32x16)"
  timeout -k 10 200 python3 bench.py $ARGS > gpurun_out/r06_mfma16_enc_A$i.json
  python3 - <<PY
import json; d=json.loads(open("gpurun_out/r06_mfma16_enc_A$i.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["conv_ms_per_step"], r["frac"], d["library"]["build"] if "library" in d else "")
PY
  echo "== PR_EXPERIMENT=16"
  POSERISK_LIB_PATH=$PWD/scripts/ab_libs/mfma16.so timeout -k 10 200 python3 bench.py $ARGS > gpurun_out/r06_mfma16_enc_B$i.json
  python3 - <<PY
import json; d=json.loads(open("gpurun_out/r06_mfma16_enc_B$i.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["conv_ms_per_step"], r["frac"], d["library"]["build"] if "library" in d else "")
PY
done
