#!/bin/bash
# LDS bank-conflict counters of one stand-alone kernel script: gpurun -- 'bash scripts/pmc_lds.sh scripts/exp_bottleneck256.py bottleneck256'
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/pmc_lds
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $ROOT/gpurun_out/pmc_lds -o pmc --output-format csv -- python3 $ROOT/$1 > /dev/null 2>&1
python3 - "$ROOT" "$2" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/gpurun_out/pmc_lds/**/*counter_collection.csv", recursive=True)[0]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if sys.argv[2] not in k and "conv" not in k and "expand" not in k:
        continue
    k = k[:70]
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in tot:
    d = {c: v / n[(k, c)] for c, v in tot[k].items()}
    print(f"{k:70s} bank conflict cycles {d.get('SQ_LDS_BANK_CONFLICT', 0):12.0f} / LDS active {d.get('SQ_LDS_IDX_ACTIVE', 0):12.0f} = {d.get('SQ_LDS_BANK_CONFLICT', 0) / max(d.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f}")
PY
