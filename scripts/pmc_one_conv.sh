#!/bin/bash
# PMC passes over one conv shape: bash scripts/pmc_one_conv.sh <name> B H Cin Cout k stride pad cfg
set -eo pipefail
NAME=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$NAME
export TMPDIR=/tmp
cd /tmp
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $SET -d "$OUT/p$i" -o pmc --output-format csv -- \
      python3 "$ROOT/scripts/one_conv.py" "$@" 10 > /dev/null 2>&1
  echo "pass $i ok"
done
