#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
python -m qserve_amd.build 2>&1 | tail -1
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_fused_gpu.py tests/test_loader.py -q -x --timeout 600 --tb=short -m gpu 2>&1 | tail -8
for v in "" "--op-by-op" ""; do
  timeout 300 python bench.py --steps 48 --no-cpu-baseline --no-prefill --no-extras $v 2>/dev/null > /tmp/ab.json
  python - "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print("variant", sys.argv[1], d["value"], d["ms_per_step"], [k["us"] for k in d["kernels"]], flush=True)
PY
done
