#!/bin/bash
# Round 6, step 1: what clock does the chip hold on the other bf16 MFMA shape?  The layer3 whole-block kernel with every
# 32x32x16 MFMA issued as two 16x16x32 ones (PR_EXPERIMENT=16: same matrix-pipe cycles, operand reads and registers; results wrong)
# against the shipped kernel, same box, A B A B.   gpurun -- 'bash scripts/exp_mfma16_shape.sh'
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for i in 1 2; do
  echo "== shipped (32x32x16)"; timeout -k 10 120 python3 scripts/exp_bottleneck256.py | cut -c1-120
  echo "== PR_EXPERIMENT=16 (two 16x16x32 per MFMA; wrong results)"
  POSERISK_LIB_PATH=$PWD/scripts/ab_libs/mfma16.so timeout -k 10 120 python3 scripts/exp_bottleneck256.py | cut -c1-120
done
