#!/bin/bash
# round 5: did the round's robustness work (bounded waits, selectable epilogue, masked epilogue-operand DMA) cost anything?  The four
# decode GEMMs, the row kernels and the attention launch the step issues under the round-4 library and under this tree, one call.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
for rep in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then export QS_AMD_LIBRARY=$ROOT/_ab_old/libqserve_amd_r4.so; else unset QS_AMD_LIBRARY; fi
    echo "--- $lib (rep $rep)"
    timeout 300 python scripts/bench_gemm_ab.py 2>&1 | grep -v amdgpu.ids
    FUSED=1 B=64 LS=1033 VARS=0 ROUNDS=5 timeout 200 python scripts/bench_attn.py 2>&1 | grep "^KV"
    B=64 LS=1033 VARS=0 ROUNDS=5 timeout 200 python scripts/bench_attn.py 2>&1 | grep "^KV"
    timeout 200 python scripts/bench_rows_ab.py 2>&1 | grep -v amdgpu.ids | tail -6
  done
done
unset QS_AMD_LIBRARY
