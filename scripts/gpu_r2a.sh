#!/bin/bash
# round 2, call A: full GPU parity suite + default bench + eager (no-graph) bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m qserve_amd.build 2>&1 | tail -1
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 --tb=short -x > gpurun_out/pytest_gpu_r2a.log 2>&1
grep -E "^(E   |FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu_r2a.log | cut -c1-300 | sort | uniq -c | head -40
echo "=== bench"
timeout 600 python bench.py --steps 64 --warmup 8 2> gpurun_out/bench_r2a.err > gpurun_out/bench_r2a.json
cut -c1-600 gpurun_out/bench_r2a.json
echo "=== bench eager"
timeout 600 python bench.py --steps 32 --warmup 4 --no-graph --no-cpu-baseline --no-kernel-bench --no-prefill 2> gpurun_out/bench_r2a_eager.err > gpurun_out/bench_r2a_eager.json
cut -c1-400 gpurun_out/bench_r2a_eager.json
