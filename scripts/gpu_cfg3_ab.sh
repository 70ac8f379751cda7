#!/bin/bash
# Round 6: BASELINE config 3's per-group decode GEMMs (g128, 128 tokens, and 64 tokens for the per-group twin of config 2) - this
# tree against a previous library (_ab_old/$OLD), alternating processes, weights from HBM, the dispatcher's own geometry.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
python -m qserve_amd.build 2>&1 | tail -1
SH="128,6144,4096 128,4096,4096 128,28672,4096 128,4096,14336 64,6144,4096 64,4096,4096 64,28672,4096 64,4096,14336"
OUT=gpurun_out/${1:-round6}_cfg3_ab.txt
{
for rep in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then export QS_AMD_LIBRARY=$ROOT/_ab_old/${OLD:-libqserve_amd_r5.so} QS_AMD_LIBRARY_AB=1; else unset QS_AMD_LIBRARY QS_AMD_LIBRARY_AB; fi
    echo "=== $lib (rep $rep), per-group"
    MODE=group VARIANTS=-1 timeout 300 python scripts/bench_gemm_shard.py $SH
  done
done
unset QS_AMD_LIBRARY QS_AMD_LIBRARY_AB
echo "=== new, per-channel (reference point)"
MODE=chn VARIANTS=-1 timeout 300 python scripts/bench_gemm_shard.py $SH
} 2>&1 | grep -v "^\[qserve\|amdgpu.ids" | tee $OUT
