#!/bin/bash
# Samples the GPU's clock / power (rocm-smi) while bench.py runs: is the fp32 step held down by the power limit?
#   gpurun -- 'bash scripts/exp_clock_sample.sh [bench args...]'
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python3 bench.py --cpu-frames 0 --no-roofline --steps 1500 --repeats 4 "$@" > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
pid=$!
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed -e 's/.*sclk clock level: [0-9S]*: //' -e 's/.*Power (W): / W /' | tr '\n' ' '
  echo
  sleep 0.7
done
wait $pid
python3 -c "
import json; d=json.loads(open('gpurun_out/clock_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['value_spread'], d.get('frames_per_s_one_batch_in_flight'))"
