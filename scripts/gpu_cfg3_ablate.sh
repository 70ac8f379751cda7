#!/bin/bash
# Round 6: where do BASELINE config 3's per-group decode GEMMs (g128, 128 tokens) lose against the per-channel ones?
# One process per line, weights from HBM (bench_gemm_shard.py), the dispatcher's own geometry:
#   chn            per-channel kernels at the same shapes
#   grp            per-group kernels (level-2 dequant in registers + the s2 scale / zero stream)
#   grp, no VALU   the same launches from the TIMING library with the level-2 arithmetic switched off (ring flag 8192: results wrong
#                  by design; the meta DMA, the LDS reads and the MFMAs stay) -> what is left above `chn` is the meta stream
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
python -m qserve_amd.build 2>&1 | tail -1
python -m qserve_amd.build --timing 2>&1 | tail -1
SH="128,6144,4096 128,4096,4096 128,28672,4096 128,4096,14336"
OUT=gpurun_out/${1:-round6}_cfg3_ablation.txt
{
echo "=== chn (product library)"
MODE=chn VARIANTS=-1 timeout 300 python scripts/bench_gemm_shard.py $SH
echo "=== grp (product library)"
MODE=group VARIANTS=-1 timeout 300 python scripts/bench_gemm_shard.py $SH
echo "=== grp (timing library, everything on)"
QS_AMD_LIBRARY=$ROOT/qserve_amd/libqserve_amd_timing.so MODE=group VARIANTS=-1 timeout 300 python scripts/bench_gemm_shard.py $SH
echo "=== grp, level-2 arithmetic OFF (timing library, ring flag 8192; results wrong by design)"
QS_AMD_LIBRARY=$ROOT/qserve_amd/libqserve_amd_timing.so RING_FLAGS=8192 MODE=group VARIANTS=-1 timeout 300 python scripts/bench_gemm_shard.py $SH
echo "=== chn / grp once more (drift check)"
MODE=chn VARIANTS=-1 timeout 300 python scripts/bench_gemm_shard.py $SH
MODE=group VARIANTS=-1 timeout 300 python scripts/bench_gemm_shard.py $SH
} 2>&1 | grep -v "^\[qserve" | tee $OUT
