#!/bin/bash
# One gpurun call of round 4.  Usage: gpu_round4.sh TAG [attn_tests|attn_ab|tests|bench|prof|pmc|gemm_ab|ref_engine ...]
#   attn_tests = the attention / race-screen / decode-step GPU tests only (fast iteration on the attention kernel)
#   attn_ab    = scripts/bench_attn.py under the OLD library (_ab_old/libqserve_amd_r3.so, built from the round-3 HEAD, git-ignored
#                but shipped by gpurun) and the NEW one, interleaved inside this one call (box-to-box spread is ~3 %)
#   tests      = whole GPU suite + smoke;  bench = default bench line;  prof = rocprofv3 kernel stats;  pmc = HBM traffic passes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-round4}
shift
WHAT=${*:-tests bench}
python -m qserve_amd.build 2>&1 | tail -1
OLD=$ROOT/_ab_old/libqserve_amd_r3.so
for w in $WHAT; do
case $w in
attn_tests)
  echo "=== attention tests"
  timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_race_screen_gpu.py tests/test_fused_gpu.py -q -m gpu --timeout 600 --tb=short > gpurun_out/pytest_attn_$TAG.log 2>&1
  grep -E "^(E   |FAILED|ERROR)|passed|failed" gpurun_out/pytest_attn_$TAG.log | cut -c1-300 | sort | uniq -c | head -30 ;;
attn_ab)
  echo "=== attention A/B (old = round-3 library, new = this tree)"
  for rep in 1 2; do
    for lib in old new; do
      if [ $lib = old ]; then export QS_AMD_LIBRARY=$OLD; else unset QS_AMD_LIBRARY; fi
      echo "--- $lib (rep $rep) B=64"
      B=64 LS=${LS:-300,640,1030,1100,1280,1535,2000,4096} VARS=0 timeout 300 python scripts/bench_attn.py 2>&1 | grep "^KV"
    done
  done
  for lib in old new; do
    if [ $lib = old ]; then export QS_AMD_LIBRARY=$OLD; else unset QS_AMD_LIBRARY; fi
    echo "--- $lib B=128"
    B=128 LS=1033,1535 VARS=0 timeout 300 python scripts/bench_attn.py 2>&1 | grep "^KV"
    echo "--- $lib B=8 (split-KV) and H=64 (G=8)"
    B=8 LS=1033,8191 VARS=0 timeout 300 python scripts/bench_attn.py 2>&1 | grep "^KV"
    B=64 H=64 LS=1033,1535 VARS=0 timeout 300 python scripts/bench_attn.py 2>&1 | grep "^KV"
  done
  unset QS_AMD_LIBRARY ;;
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
  python - <<PY
import json
for f in ("gpurun_out/bench_$TAG.json",):
    try:
        d = json.load(open(f))
        print(f, d["value"], d["ms_per_step"], [(k["kernel"].split("[")[1].split(" ")[0], k["us"]) for k in d["kernels"]])
    except Exception as e:
        print(f, "unreadable", e)
PY
  ;;
bench_ab)
  echo "=== bench.py under the old and the new library (same call)"
  for lib in old new old new; do
    if [ $lib = old ]; then export QS_AMD_LIBRARY=$OLD; else unset QS_AMD_LIBRARY; fi
    timeout 600 python bench.py --no-cpu-baseline --no-prefill --no-extras 2>/dev/null > gpurun_out/bench_${TAG}_$lib.json
    python - <<PY
import json
d = json.load(open("gpurun_out/bench_${TAG}_$lib.json"))
print("$lib", d["value"], d["ms_per_step"], [(k["kernel"].split("[")[1].split(" ")[0], k["us"]) for k in d["kernels"]])
PY
  done
  unset QS_AMD_LIBRARY ;;
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
