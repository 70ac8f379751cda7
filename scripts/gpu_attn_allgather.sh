#!/bin/bash
# Round 6: attention + quant hand-over, payload -> finisher (variant 0) against the all-gather of the row statistics (variant 6)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python -m qserve_amd.build 2>&1 | tail -1
OUT=gpurun_out/round6_attn_allgather.txt
: > $OUT
echo "plain" | tee -a $OUT
B=64 LS=1033,1535 VARS=0 ROUNDS=5 timeout 300 python scripts/bench_attn.py 2>&1 | grep "^KV" | tee -a $OUT
echo "fused: variant 0 = payload to the finisher, 6 = all-gather" | tee -a $OUT
FUSED=1 B=64 LS=1033,1280,1535 VARS=0,6 ROUNDS=7 timeout 300 python scripts/bench_attn.py 2>&1 | grep "^KV" | tee -a $OUT
FUSED=1 B=128 LS=1033 VARS=0,6 ROUNDS=5 timeout 300 python scripts/bench_attn.py 2>&1 | grep "^KV" | tee -a $OUT
FUSED=1 B=16 LS=700 VARS=0,6 ROUNDS=5 timeout 300 python scripts/bench_attn.py 2>&1 | grep "^KV" | tee -a $OUT
