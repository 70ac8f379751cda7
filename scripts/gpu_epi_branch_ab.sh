#!/bin/bash
# Round 6: the tiled GEMM's per-channel epilogue convention as a wave-uniform branch (this tree) against the per-element select
# (_ab_old/libqserve_amd_r6sel.so = the tree one commit earlier), alternating processes in one call; also the gate_up (+ silu.mul) shape
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
python -m qserve_amd.build 2>&1 | tail -1
OUT=gpurun_out/round6_epi_branch_ab.txt
: > $OUT
for rep in 1 2 3; do
  for lib in select branch; do
    if [ $lib = select ]; then export QS_AMD_LIBRARY=$ROOT/_ab_old/libqserve_amd_r6sel.so; else unset QS_AMD_LIBRARY; fi
    echo "--- $lib (rep $rep)" | tee -a $OUT
    PERGROUP=0 timeout 300 python scripts/bench_gemm_big.py 4096x4096x4096 8192x6144x4096 8192x28672x4096 65536x4096x4096 2>&1 | grep -v amdgpu.ids | grep "per-channel" | tee -a $OUT
  done
done
unset QS_AMD_LIBRARY
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -m gpu --timeout 600 --tb=short -x 2>&1 | tail -3 | tee -a $OUT
