#!/bin/bash
# PMC passes over the fused RL kernel (run on the GPU box): tools/pmc_fused.sh <outdir>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_IFETCH SQ_INSTS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "rl_fused" --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_kernels.py --only-rl --skip-dense > $OUT/p$i.log 2>&1
done
