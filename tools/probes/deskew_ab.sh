#!/bin/bash
# A/B of two builds of liblsrecon.so on ONE box (box-to-box spread is larger than most kernel changes):
#   tools/probes/deskew_ab.sh <other.so> [rounds]
# swaps the library file in place between runs and prints the deskew timings of tools/bench_kernels.py.
set -e
cur=shrimpy_amd/csrc/liblsrecon.so
cp $cur /tmp/lsr_ab_new.so
for round in $(seq 1 ${2:-2}); do
  for which in new old; do
    if [ $which = new ]; then cp /tmp/lsr_ab_new.so $cur; else cp "$1" $cur; fi
    echo "== round $round: $which"
    python tools/bench_kernels.py --reps 10 2>/dev/null | grep '"deskew_kernel' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(f\"{d['kernel']:58s} {d.get('workload',''):8s} {d['ms']:.3f} ms\")"
  done
done
cp /tmp/lsr_ab_new.so $cur
