#!/bin/bash
# Round 6, call E: (1) per-group ring kernels with the next round's dequant pinned inside the round (this tree) against the round-5
# library, (2) down_proj K-slice geometries under the new slice -> XCD mapping, weights from HBM.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
python -m qserve_amd.build 2>&1 | tail -1
OLD=libqserve_amd_r5.so bash scripts/gpu_cfg3_ab.sh ${1:-round6_e} | tail -45
{
echo "=== down_proj M=64 (per-channel): dispatcher, <2,1> x 2 slices, <2,2> x 4 slices, <2,2> x 2, <4,2> x 4, <2,1> x 4; K slices across XCDs (default) and on one XCD (RING_FLAGS=4096)"
MODE=chn VARIANTS=-1,4221,4422,4222,4442,4421 timeout 300 python scripts/bench_gemm_shard.py 64,4096,14336 32,4096,14336 128,4096,14336
RING_FLAGS=4096 MODE=chn VARIANTS=-1,4221,4422,4222,4442,4421 timeout 300 python scripts/bench_gemm_shard.py 64,4096,14336 32,4096,14336 128,4096,14336
MODE=chn VARIANTS=-1,4221,4422,4222,4442,4421 timeout 300 python scripts/bench_gemm_shard.py 64,4096,14336
} 2>&1 | grep -v "^\[qserve\|amdgpu.ids" | tee gpurun_out/${1:-round6_e}_down_geometries.txt
