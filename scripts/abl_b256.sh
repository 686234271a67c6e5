#!/bin/bash
# Stamps of bottleneck256_bf16 (timing builds; scripts/ab_libs/timing_new.so = this tree with -DPR_TIMING_HOOKS, timing_old.so =
# the commit before the MFMA-shape conversion, built the same way):   gpurun -- 'bash scripts/abl_b256.sh'
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
for which in old new old new; do
  echo "== $which"
  POSERISK_LIB_PATH=$PWD/scripts/ab_libs/timing_$which.so POSERISK_B256_STAMPS=gpurun_out/b256_stamps_$which.bin timeout -k 10 120 python3 scripts/exp_bottleneck256.py | cut -c1-110
  python3 scripts/b256_stamps.py gpurun_out/b256_stamps_$which.bin
done
