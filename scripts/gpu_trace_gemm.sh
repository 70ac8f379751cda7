#!/bin/bash
# ring GEMM timelines of the four decode shapes (library: QS_EXTRA_HIPCC_FLAGS=-DQS_RING_TRACE python -m qserve_amd.build --timing --force)
# env SHAPES="NxK ..." (default: the four GEMMs of a Llama-3-8B layer), M (default 64)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export QS_AMD_LIBRARY=$(pwd)/qserve_amd/libqserve_amd_timing.so
for shape in ${SHAPES:-6144x4096 4096x4096 28672x4096 4096x14336}; do
  M=${M:-64} N=${shape%x*} K=${shape#*x} timeout 120 python scripts/trace_gemm.py 2>&1 | grep -v amdgpu.ids
done
