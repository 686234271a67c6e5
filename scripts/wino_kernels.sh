#!/bin/bash
# Per-kernel durations of the three Winograd layer shapes (input transform | grouped GEMM | output transform), stand-alone:
#   gpurun -- 'bash scripts/wino_kernels.sh'            (extra environment for the runs in ENVV)
set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for sh in "64 28 128 128" "64 14 256 256" "64 7 512 512"; do
  set -- $sh
  d=gpurun_out/wk_$2
  rm -rf $d
  env $ENVV timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $d -o w --output-format csv -- python3 scripts/one_conv.py $sh 3 1 1 -5 40 > $d.log 2>&1 </dev/null
  f=$(find $d -name "*kernel_stats.csv" | sed -n 1p)
  echo "== $sh  $(tail -1 $d.log)"
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "pr::" in n:
        print(f"   {float(r['AverageNs']) / 1e3:8.1f} us x {r['Calls']:>4s}  {n.split('pr::(anonymous namespace)::')[-1][:70]}")
PY
  fi
done
