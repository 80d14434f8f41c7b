#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and rocprofv3 stats of the affine kernels under the
# blending border rule at config-3 size (run on the GPU box): tools/pmc_affine_grid.sh <outdir>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
for which in planar both; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "affine" --output-format csv -d $OUT/$which/$c -- python3 $R/tools/run_affine.py $which all grid-constant > $OUT/${which}_$c.log 2>&1
  done
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$which/stats -- python3 $R/tools/run_affine.py $which all grid-constant > $OUT/${which}_stats.log 2>&1
done
cd $R && python3 tools/pmc_summary.py $OUT/planar $OUT/both > $OUT/summary.txt
