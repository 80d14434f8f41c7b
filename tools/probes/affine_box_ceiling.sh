#!/bin/bash
# Where the LDS-staged affine kernel's time goes at config-3 size (probe build of csrc/affine_box.hip, -DLSR_BOX_PROBES):
#   LSR_BOX_PROBE=0 the kernel; 1 no staging (arithmetic + LDS reads + stores on whatever LDS holds); 3 staging + stores only.
# plus the PMC traffic of the real kernel.  bash tools/probes/affine_box_ceiling.sh <tag>
set -e
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-rXX}_affine_box_ceiling.txt
: > $out
for p in 0 1 3; do
  echo "probe $p" >> $out
  LSR_BOX_PROBE=$p LSR_LIBRARY=$R/shrimpy_amd/csrc/liblsrecon_boxprobe.alt timeout -k 10 200 python3 $R/tools/bench_kernels.py --only-affine --reps 5 2>/dev/null | python3 -c "
import json, sys
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    if d.get('path') == 'box': print('  %-62s %.3f ms  %.3f of 8 TB/s' % (d['kernel'], d['ms'], d['frac_of_8TBps']))" >> $out
done
cat $out
