#!/bin/bash
# Profiles of the benchmark, run on the GPU box:
#   gpurun -- 'bash scripts/profile_round.sh r03'            configs[1]: B=64 fp32 (the headline)
#   gpurun -- 'bash scripts/profile_round.sh r03 bf16'       configs[2]: B=256 bf16 encoder
# Writes under gpurun_out/<tag>_*; scripts/pmc_summary.py <tag> [bf16] turns the PMC passes into
# profiles/<tag>_hbm_traffic_*.json and profiles/<tag>_pmc_mfma_busy_*.txt and copies the kernel statistics.
# Counters are collected in their own passes (no tracing flags beside --pmc).
set -eo pipefail
TAG=${1:-r02}
MODE=${2:-fp32}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
if [ "$MODE" = bf16 ]; then
  ARGS="--precision bf16 --batch 256"; LANES=2; SFX=b256_bf16
else
  ARGS=""; LANES=3; SFX=b64
fi
cd "$ROOT"
timeout -k 10 400 python3 bench.py $ARGS --lanes $LANES > "$OUT/${TAG}_bench_${SFX}_default.json"
echo "[1/5] default bench done"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_ktrace_${SFX}" -o kt --output-format csv -- \
    python3 "$ROOT/bench.py" $ARGS --lanes 1 --cpu-frames 0 --no-other-configs --steps 20 > "$OUT/${TAG}_bench_${SFX}_lanes1_under_rocprof.json"
echo "[2/5] kernel trace done"
# the headline mode itself ($LANES batches in flight): per-kernel start/end times for the overlap summary
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_ktrace_lanes_${SFX}" -o kt --output-format csv -- \
    python3 "$ROOT/bench.py" $ARGS --lanes $LANES --cpu-frames 0 --no-roofline --no-other-configs --steps 30 --repeats 1 > "$OUT/${TAG}_bench_${SFX}_lanes_under_rocprof.json"
echo "[2b] kernel trace of the headline mode done"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C -d "$OUT/${TAG}_pmc_${SFX}_$C" -o pmc --output-format csv -- \
      python3 "$ROOT/bench.py" $ARGS --lanes 1 --cpu-frames 0 --no-roofline --no-other-configs --steps 4 --warmup 2 > /dev/null
  echo "[pmc] $C done"
done
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/${TAG}_pmc_${SFX}_MFMA" -o pmc --output-format csv -- \
    python3 "$ROOT/bench.py" $ARGS --lanes 1 --cpu-frames 0 --no-roofline --no-other-configs --steps 4 --warmup 2 > /dev/null
echo "[5/5] MFMA busy done"
find "$OUT" -name "*.csv" -path "*${TAG}_*${SFX}*" | head -30
