#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
QS_EXTRA_HIPCC_FLAGS=-DQS_RING_TRACE python -m qserve_amd.build --timing --force 2>&1 | tail -1
export QS_AMD_LIBRARY=$(pwd)/qserve_amd/libqserve_amd_timing.so
N=6144 SILU=0 timeout 120 python scripts/trace_heads.py 2>&1 | grep -v amdgpu.ids
N=28672 SILU=1 timeout 120 python scripts/trace_heads.py 2>&1 | grep -v amdgpu.ids
unset QS_AMD_LIBRARY
python -m qserve_amd.build --timing --force 2>&1 | tail -1
python scripts/bench_heads.py 2>&1 | grep launches
