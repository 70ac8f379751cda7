#!/bin/bash
# HBM traffic of the dominant kernels from the L2 memory-side counters: two separate rocprofv3 --pmc passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").  Usage: gpu_pmc.sh TAG
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r01}
python -m qserve_amd.build 2>&1 | tail -1
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$C -o pmc -- python $ROOT/scripts/pmc_workload.py > /tmp/pmc_${TAG}_$C.log 2>&1 )
  tail -2 /tmp/pmc_${TAG}_$C.log
  f=$(find /tmp/pmc_${TAG}_$C -name "*counter_collection*.csv" | head -1)
  echo "counter file: $f"
  [ -n "$f" ] && python scripts/pmc_summarise.py "$f" $C > gpurun_out/pmc_${TAG}_$C.json && cat gpurun_out/pmc_${TAG}_$C.json
done
