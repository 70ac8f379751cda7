#!/bin/bash
# round 5: the fitted split-KV model (variant 0) against forced factors, one process per batch size
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for kv in "" "--kv8"; do
for B in 1 4 8 16 24 32 48; do
  B=$B LS=1030,2048,4096,7700 VARS=0,101,102,103,104,106,108 NL=8 ROUNDS=3 timeout 280 python scripts/bench_attn.py $kv 2>/dev/null | grep "L=" | sed -e 's/variant //g' -e 's/ *[0-9]* GB\/s//g'
done
done
