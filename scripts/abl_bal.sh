#!/bin/bash
# Ablations of the evenly dealt bf16 convolution (timing build: POSERISK_CXXFLAGS=-DPR_TIMING_HOOKS python -m poserisk_release_amd.build --force)
cd "${GRAFT_REPO_ROOT:-.}"
for d in ${BAL_DBGS:-0 1 2 4 5 6 8 13}; do echo "== dbg=$d"; POSERISK_BAL_DBG=$d timeout -k 10 120 python3 scripts/exp_bal.py 256 "${BAL_ONLY:-}" 2>&1 | grep "layer\|rror"; done
