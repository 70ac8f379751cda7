#!/bin/bash
# In-run A/B of bench.py under GEMM variants (box-to-box variance is ~5-10 %, so compare inside one gpurun call).
# usage: ab_bench.sh "4001 -1 4001 -1" [extra bench.py args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
VARS=${1:-"-1"}
shift
for v in $VARS; do
  timeout 300 python bench.py --steps 64 --no-cpu-baseline --no-prefill --gemm-variant $v "$@" 2>/dev/null > /tmp/ab_$$.json
  python - "$v" /tmp/ab_$$.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print("variant", sys.argv[1], d["value"], d["ms_per_step"], [k["us"] for k in d["kernels"]], flush=True)
PY
done
