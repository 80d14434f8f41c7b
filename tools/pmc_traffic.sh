#!/bin/bash
# HBM-side traffic of the RL kernels (run on the GPU box): tools/pmc_traffic.sh <outdir> [bench args]
# LSR_TRAFFIC_REGEX: the kernels to count; LSR_TRAFFIC_PROGRAM: a python script to profile instead of bench.py (its
# arguments follow <outdir>; the bench flags below are then left out)
# Two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), then tools/traffic_stamp.py writes the stamped
# record (kernel symbol + kernel-source fingerprint) next to the CSVs.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
shift
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  if [ -n "$LSR_TRAFFIC_PROGRAM" ]; then
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "${LSR_TRAFFIC_REGEX:-.}" --output-format csv -d $OUT/$c -- python3 $R/$LSR_TRAFFIC_PROGRAM "$@" > $OUT/$c.log 2>&1
  else
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "${LSR_TRAFFIC_REGEX:-rl_fused|correlate_sep|correlate_dense|deskew_kernel}" --output-format csv -d $OUT/$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/$c.log 2>&1
  fi
done
cd $R && python3 tools/traffic_stamp.py $OUT --key ${LSR_TRAFFIC_KEY:-fused} --workload ${LSR_TRAFFIC_WORKLOAD:-config2} --kernel ${LSR_TRAFFIC_KERNEL:-rl_fused_sep_kernel}
