#!/bin/bash
# Where the time of affine_box_kernel goes (run on the GPU box, library built with
# `make -C shrimpy_amd/csrc EXTRA=-DLSR_BOX_PROBES`): 0 = the kernel, 1 = no staging (arithmetic and
# stores only), 3 = staging and stores only.
for p in 0 1 3; do echo "probe $p"; LSR_BOX_PROBE=$p timeout -k 10 200 python tools/bench_kernels.py --only-affine 2>&1 | grep '"path": "box"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(' ', d['kernel'], round(d['ms'],3))"; done
