#!/bin/bash
# round 5: 128-token ring workgroups (8 m-tiles) against the 64-token geometries at M = 96 / 128, per-channel and g128, weights from HBM
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
V=${VARIANTS:--1,4004,4182,4184,4282,4284,4482,4144,4142,4141}
for mode in group chn; do
  echo "=== MODE=$mode (variants $V)"
  MODE=$mode VARIANTS=$V timeout 600 python scripts/bench_gemm_shard.py ${SHAPES:-128,6144,4096 128,4096,4096 128,28672,4096 128,4096,14336 96,28672,4096} 2>&1 | grep "^M="
done
