#!/bin/bash
# lanes re-measured on the round's final library: B = 64 fp32 with 2 / 3 / 4 batches in flight, bf16 B = 256 with 2 / 3
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do
for l in 3 2 4 5; do
  timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-other-configs --cpu-frames 0 --repeats 3 --no-roofline --lanes $l > gpurun_out/lanes_$l.json
  python3 -c "import json;d=json.loads(open('gpurun_out/lanes_$l.json').read().strip().splitlines()[-1]);print('fp32 B=64 lanes $l:', round(d['value']), d['value_spread']['min'], d['value_spread']['max'])"
done
for l in 2 3; do
  timeout -k 10 200 python3 bench.py --precision bf16 --batch 256 --steps 60 --warmup 10 --no-other-configs --cpu-frames 0 --repeats 3 --no-roofline --lanes $l > gpurun_out/lanes_bf16_$l.json
  python3 -c "import json;d=json.loads(open('gpurun_out/lanes_bf16_$l.json').read().strip().splitlines()[-1]);print('bf16 B=256 lanes $l:', round(d['value']), d['value_spread']['min'], d['value_spread']['max'])"
done
done
