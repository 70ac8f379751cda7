#!/bin/bash
# round 2, call B: new GPU tests (loader TP, bench TP flow), the new bench line, rocprofv3 kernel trace
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m qserve_amd.build 2>&1 | tail -1
timeout 900 python -m pytest tests -q -m gpu --timeout 600 --tb=short -x -k "loader or bench_tp or fused" > gpurun_out/pytest_gpu_r2b.log 2>&1
grep -E "^(E   |FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu_r2b.log | cut -c1-300 | sort | uniq -c | head -40
echo "=== bench"
timeout 900 python bench.py 2> gpurun_out/bench_r2b.err > gpurun_out/bench_r2b.json
tail -3 gpurun_out/bench_r2b.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r2b.json"))
print(d["value"], d["ms_per_step"])
print({k:v for k,v in d["config"].items() if k not in ("workload","op_sequence","e2e_note")})
print(d["roofline"]); print(d["roofline_family"]); print(d["cpu_baseline"])
for k in d["kernels"]: print(k)
PY
echo "=== self-launch --gpus 2 on a 1-GPU box must fail loudly, not hang"
QS_DIST_BACKEND=gloo QS_DIST_DEVICE=0 timeout 300 python bench.py --gpus 2 --model tiny --batch 4 --prompt-len 96 --max-new 48 --steps 4 --warmup 2 --no-prefill 2>&1 | tail -2 | cut -c1-400
echo "=== rocprofv3 kernel trace"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r2b -o trace -- python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-prefill > /tmp/prof_r2b.log 2>&1 )
for f in $(find /tmp/prof_r2b -name "*kernel_stats*.csv" | head -1); do cp "$f" gpurun_out/round2_b_kernel_stats.csv; head -14 "$f" | cut -c1-200; done
tail -2 /tmp/prof_r2b.log | cut -c1-300
