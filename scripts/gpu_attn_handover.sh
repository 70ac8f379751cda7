#!/bin/bash
# Round 6: what the attention + quant hand-over costs, part by part (timing library; 256 / 768 compute WRONG rows by design).
#   plain            qs_single_query_attention alone
#   fused            the launch of the decode step (finisher waits for the other KV heads' granules)
#   fused, no wait   kflags 256: the finisher takes its first poll
#   fused, no wait, no publish   kflags 768: nobody writes granules either
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python -m qserve_amd.build 2>&1 | tail -1
python -m qserve_amd.build --timing 2>&1 | tail -1
export QS_AMD_LIBRARY=$(pwd)/qserve_amd/libqserve_amd_timing.so
OUT=gpurun_out/round6_attn_handover.txt
: > $OUT
for rep in 1 2; do
  echo "--- pass $rep" | tee -a $OUT
  echo "plain" | tee -a $OUT
  B=64 LS=1033,1535 VARS=0 ROUNDS=5 timeout 300 python scripts/bench_attn.py 2>&1 | grep "^KV" | tee -a $OUT
  for kf in 0 256 768; do
    echo "fused, QS_ATTN_KFLAGS=$kf" | tee -a $OUT
    QS_ATTN_KFLAGS=$kf FUSED=1 B=64 LS=1033,1535 VARS=0 ROUNDS=5 timeout 300 python scripts/bench_attn.py 2>&1 | grep "^KV" | tee -a $OUT
  done
done
