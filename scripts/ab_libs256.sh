#!/bin/bash
# Same-box A/B of BUILDS (scripts/ab_libs/<name>.so) on the stand-alone layer3 block: gpurun -- 'bash scripts/ab_libs256.sh V0 V1'
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
for round in 1 2 3; do
  for name in "$@"; do
    export POSERISK_LIB_PATH=$PWD/scripts/ab_libs/$name.so    # selected, never copied over the shipped library
    echo -n "$name: "; python3 scripts/exp_bottleneck256.py 2>/dev/null | cut -c1-100
  done
done
