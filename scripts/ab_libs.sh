#!/bin/bash
# Same-box A/B of BUILDS of the library (scripts/ab_libs/<name>.so, made here with `python -m poserisk_release_amd.build --out scripts/ab_libs/<name>.so` and shipped with the snapshot; selected with POSERISK_LIB_PATH): per build the
# stand-alone layer2 block, the in-encoder layer table rows of layer2 and the two-lane bench.
#   gpurun -- 'bash scripts/ab_libs.sh A C'
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
for round in 1 2; do
  for name in "$@"; do
    export POSERISK_LIB_PATH=$PWD/scripts/ab_libs/$name.so    # selected, never copied over the shipped library
    echo "== build $name (round $round)"
    python3 scripts/exp_bottleneck128.py 2>/dev/null | cut -c1-60
    python3 scripts/layer_table.py 256 bf16 2>/dev/null | grep -E "^L(17|20|23) |total"
    timeout -k 10 200 python3 bench.py --precision bf16 --batch 256 --lanes 2 --cpu-frames 0 --no-roofline --steps 30 --repeats 3 > gpurun_out/ab_lib.json 2>/dev/null
    python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/ab_lib.json").read().strip().splitlines()[-1])
s = d["value_spread"]
print(f"bench {d['value']:9.1f} frames/s  ({s['min']:.0f} - {s['max']:.0f})")
PY
  done
done
