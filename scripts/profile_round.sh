#!/bin/bash
# Profiles of the default benchmark, run on the GPU box:  gpurun -- 'bash scripts/profile_round.sh r01'
# Writes under gpurun_out/<tag>_*; scripts/hbm_traffic.py turns the PMC passes into profiles/<tag>_hbm_traffic_b64.json.
set -eo pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
cd "$ROOT"
timeout -k 10 400 python3 bench.py > "$OUT/${TAG}_bench_b64_default.json"
echo "[1/5] default bench done"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_ktrace" -o kt --output-format csv -- \
    python3 "$ROOT/bench.py" --lanes 1 --cpu-frames 0 --steps 20 > "$OUT/${TAG}_bench_b64_lanes1_under_rocprof.json"
echo "[2/5] kernel trace done"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C -d "$OUT/${TAG}_pmc_$C" -o pmc --output-format csv -- \
      python3 "$ROOT/bench.py" --lanes 1 --cpu-frames 0 --no-roofline --steps 4 --warmup 2 > /dev/null
  echo "[pmc] $C done"
done
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/${TAG}_pmc_MFMA" -o pmc --output-format csv -- \
    python3 "$ROOT/bench.py" --lanes 1 --cpu-frames 0 --no-roofline --steps 4 --warmup 2 > /dev/null
echo "[5/5] MFMA busy done"
find "$OUT" -name "*.csv" -path "*${TAG}_*" | head -30
