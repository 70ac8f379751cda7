#!/bin/bash
# round 5: split-KV count sweep of the matrix-core decode attention (is "aim at >= 512 workgroups" right at long contexts?)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out/round5_split_sweep.txt
: > $OUT
for kv in "" "--kv8"; do
for B in 1 2 4 8 16 32; do
  B=$B LS=${LS:-1030,2048,4096,7700} VARS=${VARS:-0,101,102,103,104,105,106,108,112,116,124,132} NL=8 ROUNDS=3 timeout 280 python scripts/bench_attn.py $kv 2>/dev/null | grep "L=" >> $OUT
done
done
cat $OUT
