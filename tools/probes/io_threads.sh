#!/bin/bash
# Streamed config-4 run under different reader,writer thread budgets (tools/stream_timeline.py).
set -e
out=${1:-gpurun_out/io_threads.jsonl}
: > "$out"
for rw in 16,16 8,8 6,9 5,10 4,11 4,8 3,6 8,16 2,6; do
  LSR_IO_THREADS=$rw timeout -k 10 200 python tools/stream_timeline.py --units 12 --scratch /dev/shm 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); d['threads']='$rw'; print(json.dumps(d))" >> "$out"
  tail -1 "$out" | cut -c1-200
done
