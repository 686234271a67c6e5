#!/bin/bash
# Stamps of bottleneck256_bf16 (timing build): gpurun -- 'bash scripts/abl_b256.sh'
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
POSERISK_CXXFLAGS=-DPR_TIMING_HOOKS python3 -m poserisk_release_amd.build --force > gpurun_out/abl_b256_build.log 2>&1
POSERISK_B256_STAMPS=gpurun_out/b256_stamps.bin timeout -k 10 120 python3 scripts/exp_bottleneck256.py | cut -c1-160
python3 scripts/b256_stamps.py gpurun_out/b256_stamps.bin
