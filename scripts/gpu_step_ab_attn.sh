#!/bin/bash
# Round 6: the decode step with the attention + quant hand-over as statistics all-gather (default) against the payload form
# (--attn-variant 7), alternating processes inside one call; BASELINE config 2 (bs = 64) and config 3 (g128, bs = 128)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python -m qserve_amd.build 2>&1 | tail -1
OUT=gpurun_out/round6_step_ab_attn.txt
: > $OUT
one() {   # $1 label, rest: bench args
  local label=$1; shift
  timeout 600 python bench.py --no-extras --no-cpu-baseline --no-prefill --no-kernel-bench --steps 64 --warmup 8 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', d['value'], 'tokens/s', d['ms_per_step'], 'ms/step')" | tee -a $OUT
}
for rep in 1 2 3; do
  one "config2 all-gather  " 
  one "config2 payload     " --attn-variant 7
done
for rep in 1 2; do
  one "config3 all-gather  " --config 3
  one "config3 payload     " --config 3 --attn-variant 7
done
