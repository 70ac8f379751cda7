#!/bin/bash
# old vs new dispatcher plan, shape by shape (scripts/bench_gemm_shard.py): lines "M,N,K old new [more]" on stdin; MODE=group for g128
cd "${GRAFT_REPO_ROOT:-/root/repo}"
while read shape old new rest; do
  [ -z "$shape" ] && continue
  VARIANTS=$old,$new${rest:+,$rest} timeout 200 python scripts/bench_gemm_shard.py $shape 2>&1 | grep -v amdgpu.ids | tail -1
done
