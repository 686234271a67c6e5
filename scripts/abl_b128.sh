#!/bin/bash
# Ablations of bottleneck128_bf16 (timing build): gpurun -- 'bash scripts/abl_b128.sh'
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p scripts/ab_libs
POSERISK_CXXFLAGS=-DPR_TIMING_HOOKS python3 -m poserisk_release_amd.build --out scripts/ab_libs/timing_hooks.so > gpurun_out/abl_b128_build.log 2>&1
export POSERISK_LIB_PATH=$PWD/scripts/ab_libs/timing_hooks.so   # the shipped library stays as it is
for d in ${DBG:-0 1 2 3 4 8 16 28 31}; do
  echo "== dbg $d"
  POSERISK_B128_DBG=$d POSERISK_B128_STAMPS=gpurun_out/b128_stamps_$d.bin timeout -k 10 120 python3 scripts/exp_bottleneck128.py | cut -c1-60
  python3 scripts/b128_stamps.py gpurun_out/b128_stamps_$d.bin
done
