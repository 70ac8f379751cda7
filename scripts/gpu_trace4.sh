#!/bin/bash
# timeline traces of the KV4 attention kernel, old (round-3 timing library) and new, inside one call; plus the memory-only / no-phase-A ablations
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
for L in ${TRACE_LS:-1033 1535}; do
  echo "##### NEW L=$L"
  QS_AMD_LIBRARY=$ROOT/qserve_amd/libqserve_amd_timing.so L=$L VAR=232 timeout 300 python scripts/trace_attn.py 2>&1 | tail -20
  echo "##### OLD L=$L"
  QS_AMD_LIBRARY=$ROOT/_ab_old/libqserve_amd_r3_timing.so L=$L VAR=232 timeout 300 python scripts/trace_attn.py 2>&1 | tail -20
done
echo "##### ablations (new timing library): 0 full, 202 no compute, 204 no phase A, 206 neither"
QS_AMD_LIBRARY=$ROOT/qserve_amd/libqserve_amd_timing.so B=64 LS=1030,1535 VARS=0,202,204,206,0 timeout 300 python scripts/bench_attn.py 2>&1 | grep "^KV"
echo "##### ablations (old timing library)"
QS_AMD_LIBRARY=$ROOT/_ab_old/libqserve_amd_r3_timing.so B=64 LS=1030,1535 VARS=0,202,204,206,0 timeout 300 python scripts/bench_attn.py 2>&1 | grep "^KV"
