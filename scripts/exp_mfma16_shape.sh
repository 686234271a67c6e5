#!/bin/bash
# Round 6: the whole-block kernels on v_mfma_f32_16x16x32_bf16 (shipped) against the build of the commit before the
# conversion (scripts/ab_libs/old_shape.so: every kernel on 32x32x16), stand-alone, same box, A B A B.
#   gpurun -- 'bash scripts/exp_mfma16_shape.sh [256|128|64]'
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
case "${1:-256}" in 256) S=scripts/exp_bottleneck256.py;; 128) S=scripts/exp_bottleneck128.py;; *) S=scripts/exp_bottleneck.py;; esac
for i in 1 2; do
  echo "== old shape (32x32x16)"; POSERISK_LIB_PATH=$PWD/scripts/ab_libs/old_shape.so timeout -k 10 120 python3 $S | cut -c1-200
  echo "== shipped (16x16x32)"; timeout -k 10 120 python3 $S | cut -c1-200
done
