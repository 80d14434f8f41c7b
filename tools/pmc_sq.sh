#!/bin/bash
# SQ counter passes over kernels matching a regex (run on the GPU box):
#   tools/pmc_sq.sh <outdir> <kernel-regex> <python script and args, relative to the repo root>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
REGEX=$2
shift 2
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"; do
  i=$((i+1))
  script=$1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "$REGEX" --output-format csv -d $OUT/p$i -- python3 $R/$script "${@:2}" > $OUT/p$i.log 2>&1
done
