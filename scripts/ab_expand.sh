#!/bin/bash
# Same-box A/B of the register-resident expansion kernel (bf16 layer2 conv3): gpurun -- 'bash scripts/ab_expand.sh'
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_hip_parity.py -x -q -m gpu -k "expand_res or bf16" > $OUT/ab_expand_tests.txt 2>&1 || { tail -30 $OUT/ab_expand_tests.txt; exit 1; }
tail -3 $OUT/ab_expand_tests.txt
for v in 1 0 1 0; do
  POSERISK_EXPAND_REGS=$v timeout -k 10 200 python3 bench.py --precision bf16 --batch 256 --lanes 2 --cpu-frames 0 --no-roofline --steps 30 --repeats 3 > $OUT/ab_expand_$v.json
  python3 - <<PY
import json; d=json.loads(open("$OUT/ab_expand_$v.json").read().strip().splitlines()[-1]); print("expand_regs=$v", d["value"], d.get("value_spread"))
PY
done
