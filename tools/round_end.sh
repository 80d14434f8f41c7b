#!/bin/bash
# The measurements a round's record is made of, in one GPU-box call (run from the repo root on the box):
#   bash tools/round_end.sh r02
# 1. PMC traffic of the fused RL launch (two --pmc passes) -> gpurun_out/<tag>_pmc/{traffic_fused.json,pmc_fused.csv}
# 2. rocprofv3 --kernel-trace --stats of the benchmark command  -> gpurun_out/<tag>_stats/
# 3. the benchmark lines themselves (config 2 with the CPU baseline, configs 4 and 5)
set -e
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
bash $R/tools/pmc_traffic.sh ${tag}_pmc
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- python3 $R/bench.py --steps 20 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_stats.log 2>&1
cd $R
# the stamped record in place so that the lines below carry `traffic`
python3 tools/traffic_stamp.py gpurun_out/${tag}_pmc --key fused --workload config2 --source profiles/${tag}_pmc_traffic_fused.csv --merge profiles/traffic.json > /dev/null
timeout -k 10 400 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
timeout -k 10 300 python3 bench.py --workload config4 --scratch /dev/shm --no-cpu-baseline > gpurun_out/${tag}_bench_config4.json 2>> gpurun_out/${tag}_bench.err
timeout -k 10 300 python3 bench.py --workload config5 --scratch /dev/shm --no-cpu-baseline > gpurun_out/${tag}_bench_config5.json 2>> gpurun_out/${tag}_bench.err
cp profiles/traffic.json gpurun_out/${tag}_traffic.json
tail -c 600 gpurun_out/${tag}_bench.json
