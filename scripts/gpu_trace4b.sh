#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
for KF in ${KFS:-0 0 2}; do
  echo "##### NEW kflags=$KF L=${L:-1033}"
  QS_ATTN_KFLAGS=$KF QS_AMD_LIBRARY=$ROOT/qserve_amd/libqserve_amd_timing.so L=${L:-1033} VAR=232 timeout 300 python scripts/trace_attn.py 2>&1 | grep -v Warning | grep -v "print(" | tail -21
done
