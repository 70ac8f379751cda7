#!/bin/bash
# One gpurun call of round 6.  Usage: gpu_round6.sh TAG [mb5 tests tests_record wide_ab bench prof pmc pmc_mfma <script> ...]
#   mb5          scripts/microbench_mfma5.hip (asm MFMAs on fixed AGPRs, one wave per SIMD)
#   tests        whole GPU suite + smoke; tests_record = the same with QS_PARITY_RECORD=1 (attention parity exceptions recorded)
#   gemm_tests   tests/test_gemm_gpu.py + bounded waits + race screen only
#   wide_ab      scripts/bench_wide_ab.py: eight-wave vs four-wave compute-bound tile, alternating in one process
#   bench / prof / pmc   bench line, rocprofv3 kernel stats of the same command, HBM traffic passes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-round6}
shift
WHAT=${*:-tests bench}
python -m qserve_amd.build 2>&1 | tail -1
for w in $WHAT; do
case $w in
mb5)
  echo "=== microbench_mfma5"
  hipcc -O3 --offload-arch=gfx950 scripts/microbench_mfma5.hip -o /tmp/mb_mfma5 2>/dev/null && timeout 120 /tmp/mb_mfma5 | tee gpurun_out/${TAG}_mb_mfma5.txt ;;
mbvalu)
  echo "=== microbench_valu"
  hipcc -O3 --offload-arch=gfx950 -Wno-unused-value scripts/microbench_valu.hip -o /tmp/mb_valu 2>/dev/null && timeout 120 /tmp/mb_valu | tee gpurun_out/${TAG}_mb_valu.txt ;;
tests|tests_record)
  echo "=== pytest -m gpu ($w)"
  if [ $w = tests_record ]; then export QS_PARITY_RECORD=1; fi
  timeout 1500 python -m pytest tests -q -m gpu --timeout 600 --tb=short > gpurun_out/pytest_gpu_$TAG.log 2>&1
  unset QS_PARITY_RECORD
  grep -E "^(E   |FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu_$TAG.log | cut -c1-300 | sort | uniq -c | head -40
  echo "=== smoke"
  timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 ;;
mbclock)
  echo "=== microbench_clock (what s_memtime counts)"
  hipcc -O3 --offload-arch=gfx950 scripts/microbench_clock.hip -o /tmp/mb_clock 2>/dev/null && timeout 120 /tmp/mb_clock | tee gpurun_out/${TAG}_mb_clock.txt ;;
pmc_ab)
  # HBM traffic with the K slices across the XCDs (default) and on one XCD (variant 9096 = the mapping of rounds 3-5)
  echo "=== PMC, K slices across the XCDs"
  bash scripts/gpu_pmc.sh ${TAG}_kxcd 2>&1 | tail -3
  echo "=== PMC, K slices on one XCD"
  QS_GEMM_VARIANT=9096 bash scripts/gpu_pmc.sh ${TAG}_onexcd 2>&1 | tail -3 ;;
attn_record)
  # step 1 of scripts/record_attention_exceptions.py: record every element beyond the attention contract (nothing asserted on them)
  echo "=== attention parity exceptions (record)"
  QS_PARITY_RECORD=1 timeout 1200 python -m pytest tests/test_attention_gpu.py -q -m gpu --timeout 600 --tb=short > gpurun_out/pytest_attn_record_$TAG.log 2>&1
  tail -2 gpurun_out/pytest_attn_record_$TAG.log
  python scripts/record_attention_exceptions.py 2>&1 | tail -14
  cp tests/golden/attention_parity_exceptions.json gpurun_out/attention_parity_exceptions_golden_$TAG.json ;;
gemm_tests)
  echo "=== GEMM tests"
  timeout 1200 python -m pytest tests/test_gemm_gpu.py tests/test_bounded_waits_gpu.py tests/test_race_screen_gpu.py tests/test_planes_gpu.py -q -m gpu --timeout 600 --tb=short > gpurun_out/pytest_gemm_$TAG.log 2>&1
  grep -E "^(E   |FAILED|ERROR)|passed|failed" gpurun_out/pytest_gemm_$TAG.log | cut -c1-300 | sort | uniq -c | head -40 ;;
wide_ab)
  echo "=== eight-wave vs four-wave tile"
  ACT=1 timeout 900 python scripts/bench_wide_ab.py ${WIDE_SHAPES:-} 2>&1 | tee gpurun_out/${TAG}_wide_ab.txt | cut -c1-260 ;;
bench)
  echo "=== bench"
  timeout 1200 python bench.py 2> gpurun_out/bench_$TAG.err > gpurun_out/bench_$TAG.json
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/bench_$TAG.json"))
    print(d["value"], d["ms_per_step"], [(k["kernel"].split("[")[1].split("]")[0], k["us"]) for k in d["kernels"]])
    print({k: d["config"].get(k) for k in ("prefill_ms", "e2e_tokens_per_s", "op_by_op_tokens_per_s", "other_configs")})
    print(d.get("roofline"), d.get("roofline_family"))
except Exception as e:
    print("bench line unreadable", e)
    print(open("gpurun_out/bench_$TAG.err").read()[-1500:])
PY
  ;;
prof)
  echo "=== rocprofv3 kernel stats (same command, shorter)"
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o trace -- python $ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extras --no-prefill > /tmp/prof_$TAG.log 2>&1 )
  for f in $(find /tmp/prof_$TAG -name "*kernel_stats*.csv" | head -1); do cp "$f" gpurun_out/${TAG}_kernel_stats.csv; head -12 "$f" | cut -c1-200; done ;;
pmc)
  echo "=== PMC"
  bash scripts/gpu_pmc.sh $TAG 2>&1 | tail -4 ;;
*)
  echo "=== custom: $w"
  if [ -f scripts/$w ]; then timeout 1200 bash scripts/$w $TAG 2>&1 | tail -60; fi ;;
esac
done
