#!/bin/bash
# Stamps of bottleneck256_bf16 (timing build): gpurun -- 'bash scripts/abl_b256.sh'
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p scripts/ab_libs
POSERISK_CXXFLAGS=-DPR_TIMING_HOOKS python3 -m poserisk_release_amd.build --out scripts/ab_libs/timing_hooks.so > gpurun_out/abl_b256_build.log 2>&1
export POSERISK_LIB_PATH=$PWD/scripts/ab_libs/timing_hooks.so   # the shipped library stays as it is
POSERISK_B256_STAMPS=gpurun_out/b256_stamps.bin timeout -k 10 120 python3 scripts/exp_bottleneck256.py | cut -c1-160
python3 scripts/b256_stamps.py gpurun_out/b256_stamps.bin
