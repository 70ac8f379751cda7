#!/bin/bash
# timing ablations of the four-wave tile (QS_TIMING library, results wrong by design): which part bounds the stage?
# 3400 + bits: 1 no MFMA, 2 no DMA, 4 no activation operand reads, 8 no barrier, 16 no weight reads / unpack
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m qserve_amd.build --timing 2>&1 | tail -1
export QS_AMD_LIBRARY=$(pwd)/qserve_amd/libqserve_amd_timing.so
python scripts/bench_wide_variant.py ${VARS:-3003,3401,3402,3404,3408,3416,3420,3406,3422,3430,3003} ${SHAPES:-4096x4096x4096 8192x4096x14336} 2>&1 | grep "M="
