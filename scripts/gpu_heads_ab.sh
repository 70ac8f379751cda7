#!/bin/bash
# round 5: the timed decode step with the row-op heads off / on (QS_HEADS), alternating inside ONE gpurun call
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2 3; do
  for h in 0 1; do
    QS_HEADS=$h timeout 600 python bench.py --no-cpu-baseline --no-prefill --no-extras --no-kernel-bench 2>/dev/null > /tmp/b.json
    python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("QS_HEADS=$h rep $rep", d["value"], d["ms_per_step"])
PY
  done
done
