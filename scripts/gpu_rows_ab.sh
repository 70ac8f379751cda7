#!/bin/bash
# row-kernel A/B: old library (env OLD under _ab_old/, default the previous build) vs this tree (scripts/bench_rows_ab.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
for rep in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then export QS_AMD_LIBRARY=$ROOT/_ab_old/${OLD:-libqserve_amd_prev.so}; else unset QS_AMD_LIBRARY; fi
    echo "--- $lib (rep $rep)"
    timeout 300 python scripts/bench_rows_ab.py 2>&1 | grep -v amdgpu.ids | tail -3
  done
done
unset QS_AMD_LIBRARY
