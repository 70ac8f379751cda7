#!/bin/bash
# Round 6: tiled GEMM with the stage stream running across tile boundaries (this tree) against _ab_old/libqserve_amd_r6sel.so
# (the tree before it), alternating processes in one call; then the GEMM suite
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
python -m qserve_amd.build 2>&1 | tail -1
OUT=gpurun_out/round6_stream_tiles_ab.txt
: > $OUT
for rep in 1 2 3; do
  for lib in stop_and_go streamed; do
    if [ $lib = stop_and_go ]; then export QS_AMD_LIBRARY=$ROOT/_ab_old/libqserve_amd_r6sel.so; else unset QS_AMD_LIBRARY; fi
    echo "--- $lib (rep $rep)" | tee -a $OUT
    timeout 300 python scripts/bench_gemm_big.py 4096x4096x4096 8192x6144x4096 8192x28672x4096 65536x4096x4096 8192x4096x14336 65536x4096x1024 2>&1 | grep -v amdgpu.ids | grep "per-channel" | tee -a $OUT
  done
done
unset QS_AMD_LIBRARY
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_race_screen_gpu.py -q -m gpu --timeout 600 --tb=short -x 2>&1 | tail -3 | tee -a $OUT
