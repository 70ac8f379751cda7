#!/bin/bash
# rocprofv3 kernel stats of the prompt phase (bench.py's end-to-end leg: 64 x 1024 prompt tokens + 511 decode steps), TAG = $1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
TAG=${1:-prefill}
export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o trace -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-bench > /tmp/prof_$TAG.log 2>&1 )
tail -2 /tmp/prof_$TAG.log | cut -c1-300
for f in $(find /tmp/prof_$TAG -name "*kernel_stats*.csv" | head -1); do cp "$f" gpurun_out/${TAG}_kernel_stats.csv; done
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))
for r in rows[:40]:
    print(r["Name"][:110], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), round(float(r["TotalDurationNs"]) / 1e6, 2))
PY
