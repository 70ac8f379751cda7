#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m qserve_amd.build 2>&1 | tail -1
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 --tb=short -x > gpurun_out/pytest_gpu_r2c.log 2>&1
grep -E "^(E   |FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu_r2c.log | cut -c1-300 | sort | uniq -c | head -20
VARS=0,201,0 LS=1033,1280,1535,4096 timeout 300 python scripts/bench_attn.py 2>&1 | grep -v amdgpu.ids
VARS=0 B=64 LS=1033,4096 timeout 300 python scripts/bench_attn.py --kv8 2>&1 | grep -v amdgpu.ids
VARS=0 B=8 LS=7680,8191 timeout 300 python scripts/bench_attn.py --kv8 2>&1 | grep -v amdgpu.ids
VARS=0 B=8 LS=7680,8191 timeout 300 python scripts/bench_attn.py 2>&1 | grep -v amdgpu.ids
echo "=== bench"
timeout 900 python bench.py --no-cpu-baseline 2> gpurun_out/bench_r2c.err > gpurun_out/bench_r2c.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r2c.json"))
print(d["value"], d["ms_per_step"])
print({k:v for k,v in d["config"].items() if k not in ("workload","op_sequence","e2e_note")})
print(d["roofline"]); print(d["roofline_family"])
for k in d["kernels"]: print(k["kernel"], k["us"], k.get("gbs"), k.get("tops"))
PY
