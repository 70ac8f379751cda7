#!/bin/bash
# One gpurun call of round 3.  Usage: gpu_round3.sh TAG [tests|bench|prof|pmc ...]   (default: tests bench)
# tests = GPU parity suite + smoke; bench = default bench line + the round-2 op sequence in the same call (A/B);
# prof = rocprofv3 kernel stats of the bench command; pmc = the two PMC passes (HBM traffic of the dominant kernels)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-round3}
shift
WHAT=${*:-tests bench}
python -m qserve_amd.build 2>&1 | tail -1
for w in $WHAT; do
case $w in
tests)
  echo "=== pytest -m gpu"
  timeout 1500 python -m pytest tests -q -m gpu --timeout 600 --tb=short > gpurun_out/pytest_gpu_$TAG.log 2>&1
  grep -E "^(E   |FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu_$TAG.log | cut -c1-300 | sort | uniq -c | head -30
  echo "=== smoke"
  timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 ;;
bench)
  echo "=== bench"
  timeout 1200 python bench.py 2> gpurun_out/bench_$TAG.err > gpurun_out/bench_$TAG.json
  cut -c1-600 gpurun_out/bench_$TAG.json
  echo "=== bench with the row-op tails (A/B in the same call)"
  timeout 600 python bench.py --tails --no-cpu-baseline --no-prefill --no-extras 2>/dev/null > gpurun_out/bench_${TAG}_tails.json
  cut -c1-300 gpurun_out/bench_${TAG}_tails.json
  python - <<PY
import json
for f in ("gpurun_out/bench_$TAG.json", "gpurun_out/bench_${TAG}_tails.json"):
    try:
        d = json.load(open(f))
        print(f, d["value"], d["ms_per_step"], [(k["kernel"].split("[")[1].split(" ")[0], k["us"]) for k in d["kernels"]])
    except Exception as e:
        print(f, "unreadable", e)
PY
  ;;
prof)
  echo "=== rocprofv3 kernel stats (same command, shorter)"
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o trace -- python $ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extras --no-prefill > /tmp/prof_$TAG.log 2>&1 )
  for f in $(find /tmp/prof_$TAG -name "*kernel_stats*.csv" | head -1); do cp "$f" gpurun_out/${TAG}_kernel_stats.csv; head -12 "$f" | cut -c1-200; done ;;
pmc)
  echo "=== PMC"
  bash scripts/gpu_pmc.sh $TAG 2>&1 | tail -4 ;;
esac
done
