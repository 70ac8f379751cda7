#!/bin/bash
# One gpurun call of round 2: GPU parity suite, smoke, the default bench line, rocprofv3 kernel stats of the same command,
# the two PMC passes (HBM traffic of the dominant kernels).  Usage: gpu_round2.sh TAG   (outputs under gpurun_out/)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-round2}
python -m qserve_amd.build 2>&1 | tail -1
echo "=== pytest -m gpu"
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 --tb=short > gpurun_out/pytest_gpu_$TAG.log 2>&1
grep -E "^(E   |FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu_$TAG.log | cut -c1-300 | sort | uniq -c | head -20
echo "=== smoke"
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3
echo "=== bench"
timeout 1200 python bench.py 2> gpurun_out/bench_$TAG.err > gpurun_out/bench_$TAG.json
cut -c1-400 gpurun_out/bench_$TAG.json
echo "=== rocprofv3 kernel stats (same command, shorter)"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o trace -- python $ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extras --no-prefill > /tmp/prof_$TAG.log 2>&1 )
for f in $(find /tmp/prof_$TAG -name "*kernel_stats*.csv" | head -1); do cp "$f" gpurun_out/${TAG}_kernel_stats.csv; head -8 "$f" | cut -c1-160; done
echo "=== PMC"
bash scripts/gpu_pmc.sh $TAG 2>&1 | tail -4
