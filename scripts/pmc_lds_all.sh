#!/bin/bash
# LDS bank-conflict share per kernel over a bench run: gpurun -- 'bash scripts/pmc_lds_all.sh' (bf16 B=256), BENCH_ARGS="--precision fp32 --batch 64" for the fp32 headline
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/pmc_lds_all
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $ROOT/gpurun_out/pmc_lds_all -o pmc --output-format csv -- python3 $ROOT/bench.py ${BENCH_ARGS:---precision bf16 --batch 256} --lanes 1 --cpu-frames 0 --no-roofline --steps 3 --warmup 1 > /dev/null 2>&1
python3 - "$ROOT" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/gpurun_out/pmc_lds_all/**/*counter_collection.csv", recursive=True)[0]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    tot[r["Kernel_Name"][:78]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0)):
    a = d.get("SQ_LDS_IDX_ACTIVE", 0)
    if a > 0:
        print(f"{k:78s} conflict / active = {d.get('SQ_LDS_BANK_CONFLICT', 0) / a:.3f}   (active {a:.3g})")
PY
