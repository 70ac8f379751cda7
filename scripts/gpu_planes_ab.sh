#!/bin/bash
# the decode step with the K-slice planes form off / on for down / on for down and o (QS_PLANES), inside one gpurun call
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do
  for p in none down down,o; do
    QS_PLANES=$([ $p = none ] && echo "" || echo $p) timeout 600 python bench.py --no-cpu-baseline --no-prefill --no-extras --no-kernel-bench 2>/dev/null > /tmp/b.json
    python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("planes=$p", "rep $rep", d["value"], d["ms_per_step"])
PY
  done
done
