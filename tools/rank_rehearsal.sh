#!/bin/bash
# Rehearsal of the sharded store-to-store run on a ONE-GPU box (verdict r4 item 2.iii): N ranks share the card (gloo carries
# the barrier and the timing reduction; the data path has no collective), each takes its round-robin share of a config-4
# plate in the acquisition's format and writes blosc-zstd frames encoded on the device.  The GPU pool allows six
# processes on a card at once, and the launcher counts as one: the rehearsal stops at five ranks (the node-wide run uses one rank per GPU).
#   bash tools/rank_rehearsal.sh <tag>       -> gpurun_out/<tag>_rank_rehearsal.jsonl
set -e
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${tag}_rank_rehearsal.jsonl
: > $out
for n in 1 2 4 5; do
  timeout -k 10 400 python3 $R/bench.py --gpus $n --workload config4 --engine-format --output-compression blosc-zstd \
      --no-cpu-baseline --steps 2 --scratch /dev/shm > $R/gpurun_out/${tag}_rehearsal_$n.json 2> $R/gpurun_out/${tag}_rehearsal_$n.err || { tail -5 $R/gpurun_out/${tag}_rehearsal_$n.err; exit 1; }
  python3 - $n $R/gpurun_out/${tag}_rehearsal_$n.json >> $out <<'PY'
import json, sys
n, path = int(sys.argv[1]), sys.argv[2]
d = json.load(open(path))
s = d["config"]["store_to_store"]
print(json.dumps({"ranks_sharing_one_gpu_and_16_cores": n, "units": s["units"], "job_seconds": round(s["seconds"], 4),
                  "aggregate_voxels_per_s": s["voxels_per_s"], "s_per_unit_aggregate": round(s["seconds"] / s["units"], 4),
                  "device_codec": s.get("device_codec"), "stage_s_per_unit_rank0": s["stage_s_per_unit"],
                  "stage_s_per_unit_by_rank": s.get("stage_s_per_unit_by_rank"),
                  "kernels_resident_ms_per_unit": d["config"]["kernels_resident"]["ms_per_unit"], "io": s["io"],
                  "parallelism": d["config"]["parallelism"]}))
PY
  tail -c 300 $out | head -c 300; echo
done
