#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ counters of the affine kernels at
# config-3 size (run on the GPU box): tools/pmc_affine.sh <outdir>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
for which in planar both; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "affine" --output-format csv -d $OUT/$which/$c -- python3 $R/tools/run_affine.py $which > $OUT/${which}_$c.log 2>&1
  done
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "affine" --output-format csv -d $OUT/$which/sq$i -- python3 $R/tools/run_affine.py $which > $OUT/${which}_sq$i.log 2>&1
  done
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$which/stats -- python3 $R/tools/run_affine.py $which > $OUT/${which}_stats.log 2>&1)
done
cd $R && python3 tools/pmc_summary.py $OUT/planar $OUT/both > $OUT/summary.txt
