#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench, rocprofv3 kernel trace.  Outputs under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r01}
echo "=== build"; python -m qserve_amd.build 2>&1 | tail -2
echo "=== pytest -m gpu"
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 2>&1 | tail -40 | tee gpurun_out/pytest_gpu_$TAG.log
echo "=== smoke"
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/smoke_$TAG.log
echo "=== bench"
timeout 1200 python bench.py --steps 64 --warmup 8 2> gpurun_out/bench_$TAG.err | tee gpurun_out/bench_$TAG.json
tail -5 gpurun_out/bench_$TAG.err
echo "=== rocprofv3"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o trace -- python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --steps 8 --warmup 2 --no-cpu-baseline > /tmp/prof_$TAG.log 2>&1 )
find /tmp/prof_$TAG -type f | head
for f in $(find /tmp/prof_$TAG -name "*kernel_stats*.csv" | head -2); do cp "$f" gpurun_out/rocprof_kernel_stats_$TAG.csv; head -25 "$f"; done
tail -3 /tmp/prof_$TAG.log
