#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
python -m qserve_amd.build 2>&1 | tail -1
for v in "" "--prefetch" "" "--prefetch"; do
  timeout 300 python bench.py --steps 48 --no-cpu-baseline --no-prefill --no-extras $v 2>/tmp/ab.err > /tmp/ab.json || tail -5 /tmp/ab.err
  python - "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print("variant", sys.argv[1], d["value"], d["ms_per_step"], [k["us"] for k in d["kernels"]], flush=True)
PY
done
