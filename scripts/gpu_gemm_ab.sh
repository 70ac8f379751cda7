#!/bin/bash
# decode GEMM A/B: old library (env OLD, default the round-3 build) vs this tree, same call, medians of interleaved rounds (scripts/bench_gemm_ab.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
for rep in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then export QS_AMD_LIBRARY=$ROOT/_ab_old/${OLD:-libqserve_amd_r3.so}; else unset QS_AMD_LIBRARY; fi
    echo "--- $lib (rep $rep)"
    timeout 300 python scripts/bench_gemm_ab.py 2>&1 | grep -v amdgpu.ids
  done
done
unset QS_AMD_LIBRARY
