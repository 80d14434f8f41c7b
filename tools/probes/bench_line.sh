#!/bin/bash
# ms per step and ms per fused RL launch of a short config-2 bench run
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['launch_ms'])"
