#!/bin/bash
# attention experiments behind qs_set_attention_variant, against the old library, one call (medians of interleaved rounds)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
VARS=${EXP_VARS:-0,3}
LSS=${LS:-640,1030,1060,1100,1150,1280,1535,2000,4096}
echo "--- new: variants $VARS"
B=64 LS=$LSS VARS=$VARS timeout 600 python scripts/bench_attn.py 2>&1 | grep "^KV" | sed 's/GB\/s//g; s/variant //g'
echo "--- old"
QS_AMD_LIBRARY=$ROOT/_ab_old/libqserve_amd_r3.so B=64 LS=$LSS VARS=0 timeout 600 python scripts/bench_attn.py 2>&1 | grep "^KV"
echo "--- B=128 / G=8 new then old"
B=128 LS=1033,1535 VARS=$VARS timeout 300 python scripts/bench_attn.py 2>&1 | grep "^KV" | sed 's/GB\/s//g; s/variant //g'
B=64 H=64 LS=1033,1535 VARS=$VARS timeout 300 python scripts/bench_attn.py 2>&1 | grep "^KV" | sed 's/GB\/s//g; s/variant //g'
QS_AMD_LIBRARY=$ROOT/_ab_old/libqserve_amd_r3.so B=128 LS=1033,1535 VARS=0 timeout 300 python scripts/bench_attn.py 2>&1 | grep "^KV"
QS_AMD_LIBRARY=$ROOT/_ab_old/libqserve_amd_r3.so B=64 H=64 LS=1033,1535 VARS=0 timeout 300 python scripts/bench_attn.py 2>&1 | grep "^KV"
