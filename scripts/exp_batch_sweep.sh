#!/bin/bash
# Per-layer conv time at several batch sizes: does a batch whose activations fit the 256 MiB Infinity Cache run faster per frame?
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
for B in 32 64 128 256; do
  echo "== bf16 B=$B"; timeout -k 10 200 python3 scripts/layer_table.py $B bf16 2>/dev/null
done
for B in 16 32 64; do
  echo "== fp32 B=$B"; timeout -k 10 200 python3 scripts/layer_table.py $B fp32 2>/dev/null
done
