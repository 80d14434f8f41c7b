#!/bin/bash
# Cache-policy probes of the fused RL kernel (run on the GPU box; rebuilds rl_fused_sep_pz9.o per variant):
#   tools/probes/fused_policy.sh
cd $GRAFT_REPO_ROOT/shrimpy_amd/csrc
run() {
  rm -f rl_fused_sep_pz9.o
  make EXTRA="$1" rl_fused_sep_pz9.o liblsrecon.so > /dev/null 2>&1 || { echo "build failed: $1"; return; }
  (cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '->', round(d['roofline']['launch_ms'],4), 'ms per launch,', round(d['ms_per_step'],2), 'ms per step')")
}
run ""
run '-DLSR_FUSED_STORE_POLICY=\"sc1\"'
run '-DLSR_FUSED_STORE_POLICY=\"nt\ sc1\"'
run '-DLSR_FUSED_STORE_POLICY=\"sc0\ sc1\"'
run '-DLSR_FUSED_GLDS_POLICY=\"nt\"'
run ""
