#!/bin/bash
# Round 6: decode step of this tree against the round-5 library (_ab_old/libqserve_amd_r5.so), alternating processes in one call
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
python -m qserve_amd.build 2>&1 | tail -1
OUT=gpurun_out/round6_step_ab_r5.txt
: > $OUT
for rep in 1 2 3; do
  for lib in round5 round6; do
    if [ $lib = round5 ]; then export QS_AMD_LIBRARY=$ROOT/_ab_old/libqserve_amd_r5.so QS_AMD_LIBRARY_AB=1; else unset QS_AMD_LIBRARY QS_AMD_LIBRARY_AB; fi
    timeout 600 python bench.py --no-extras --no-cpu-baseline --no-prefill --no-kernel-bench --steps 64 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', d['value'], 'tokens/s', d['ms_per_step'], 'ms/step')" | tee -a $OUT
  done
done
