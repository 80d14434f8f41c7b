#!/bin/bash
# The measurements a round's record is made of, in one GPU-box call (run from the repo root on the box):
#   bash tools/round_end.sh r05
# 1. PMC traffic of the fused RL launch and of the ky (x) Kzx launch (two --pmc passes each)
#        -> gpurun_out/<tag>_pmc/, <tag>_pmc_rotated/ and the stamped profiles/traffic.json
# 2. rocprofv3 --kernel-trace --stats of the benchmark command  -> gpurun_out/<tag>_stats/
# 3. the benchmark lines themselves (config 2 with the CPU baseline, configs 4 and 5 store to store in the formats of
#    DESIGN.md section 4.10, --psf rotated)
# 4. the device codecs on their own (tools/bench_codec.py) and their rocprofv3 kernel summary
set -e
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
bash $R/tools/pmc_traffic.sh ${tag}_pmc
LSR_TRAFFIC_KEY=rotated LSR_TRAFFIC_KERNEL=rl_fused_ysep_kernel bash $R/tools/pmc_traffic.sh ${tag}_pmc_rotated --psf rotated
LSR_TRAFFIC_KEY=measured LSR_TRAFFIC_KERNEL=irfft_rows_rl_kernel LSR_TRAFFIC_REGEX="rows_kernel|rows_rl_kernel|zcorr|spectrum|fft_rtc" bash $R/tools/pmc_traffic.sh ${tag}_pmc_measured --psf measured
LSR_TRAFFIC_KEY=decode LSR_TRAFFIC_KERNEL=decode_blocks_kernel LSR_TRAFFIC_REGEX="decode_blocks|unshuffle" LSR_TRAFFIC_PROGRAM=tools/bench_codec.py bash $R/tools/pmc_traffic.sh ${tag}_pmc_decode --decode --reps 2
LSR_TRAFFIC_KEY=encode LSR_TRAFFIC_KERNEL=encode_blocks_kernel LSR_TRAFFIC_REGEX="encode_blocks|scan_frames|place_frames|gather_frames" LSR_TRAFFIC_PROGRAM=tools/bench_codec.py bash $R/tools/pmc_traffic.sh ${tag}_pmc_encode --reps 2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- python3 $R/bench.py --steps 20 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_codec_decode_stats -- python3 $R/tools/bench_codec.py --decode > $R/gpurun_out/${tag}_codec_decode_stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_codec_encode_stats -- python3 $R/tools/bench_codec.py > $R/gpurun_out/${tag}_codec_encode_stats.log 2>&1
cd $R
# the stamped records in place so that the lines below carry `traffic`
python3 tools/traffic_stamp.py gpurun_out/${tag}_pmc --key fused --workload config2 --source profiles/${tag}_pmc_traffic_fused.csv --merge profiles/traffic.json > /dev/null
python3 tools/traffic_stamp.py gpurun_out/${tag}_pmc_rotated --key rotated --workload config2 --kernel rl_fused_ysep_kernel --source profiles/${tag}_pmc_traffic_rotated.csv --merge profiles/traffic.json > /dev/null
timeout -k 10 400 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
timeout -k 10 300 python3 bench.py --psf rotated --no-cpu-baseline > gpurun_out/${tag}_bench_rotated_psf.json 2>> gpurun_out/${tag}_bench.err
b4="timeout -k 10 300 python3 bench.py --workload config4 --scratch /dev/shm --no-cpu-baseline"
$b4 > gpurun_out/${tag}_bench_config4.json 2>> gpurun_out/${tag}_bench.err
$b4 --engine-format --output-compression blosc-zstd > gpurun_out/${tag}_bench_config4_engine_blosc.json 2>> gpurun_out/${tag}_bench.err
$b4 --engine-format > gpurun_out/${tag}_bench_config4_engine.json 2>> gpurun_out/${tag}_bench.err
$b4 --output-compression blosc-zstd > gpurun_out/${tag}_bench_config4_out_blosc.json 2>> gpurun_out/${tag}_bench.err
b5="timeout -k 10 300 python3 bench.py --workload config5 --scratch /dev/shm --no-cpu-baseline"
$b5 > gpurun_out/${tag}_bench_config5.json 2>> gpurun_out/${tag}_bench.err
$b5 --engine-format --output-compression blosc-zstd > gpurun_out/${tag}_bench_config5_engine_blosc.json 2>> gpurun_out/${tag}_bench.err
timeout -k 10 200 python3 tools/bench_codec.py --decode > gpurun_out/${tag}_codec_decode.json 2>> gpurun_out/${tag}_bench.err
timeout -k 10 200 python3 tools/bench_codec.py --check > gpurun_out/${tag}_codec_encode.json 2>> gpurun_out/${tag}_bench.err
timeout -k 10 200 python3 tools/bench_codec.py --from-pipeline --check > gpurun_out/${tag}_codec_encode_rl_result.json 2>> gpurun_out/${tag}_bench.err
cp profiles/traffic.json gpurun_out/${tag}_traffic.json
python3 - $tag <<'PY'
import json, sys
tag = sys.argv[1]
for name in ("bench", "bench_rotated_psf", "bench_config4", "bench_config4_engine_blosc", "bench_config4_engine", "bench_config4_out_blosc",
             "bench_config5", "bench_config5_engine_blosc"):
    d = json.load(open(f"gpurun_out/{tag}_{name}.json"))
    s = d["config"].get("store_to_store") or {}
    print(f"{name:30s} {d['ms_per_step']:8.2f} ms  {d['value']:.3e} {d['unit']}  frac {d['roofline']['frac']:.3f}"
          + (f"  store->store {s['s_per_unit']:.4f} s/unit {s.get('device_codec')}" if s else ""))
for name in ("codec_decode", "codec_encode", "codec_encode_rl_result"):
    d = json.load(open(f"gpurun_out/{tag}_{name}.json"))
    print(f"{name:30s} {d['ms']:8.3f} ms  ratio {d['ratio']}")
PY
