#!/bin/bash
# GPU parity tests only (fast iteration).  Usage: gpu_tests.sh TAG [pytest -k expr]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
TAG=${1:-t}
KEXPR=${2:-}
python -m qserve_amd.build 2>&1 | tail -1
if [ -n "$KEXPR" ]; then
  timeout 1500 python -m pytest tests -q -m gpu --timeout 600 --tb=short -k "$KEXPR" > gpurun_out/pytest_gpu_$TAG.log 2>&1
else
  timeout 1500 python -m pytest tests -q -m gpu --timeout 600 --tb=short > gpurun_out/pytest_gpu_$TAG.log 2>&1
fi
grep -E "^(E   |FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu_$TAG.log | cut -c1-300 | sort | uniq -c | head -40
