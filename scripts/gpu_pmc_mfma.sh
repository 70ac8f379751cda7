#!/bin/bash
# INT8-MFMA utilisation of the W4A8 GEMMs from the SQ counters (own rocprofv3 pass: --pmc with --kernel-trace only).
# Usage: gpu_pmc_mfma.sh TAG
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r01}
python -m qserve_amd.build 2>&1 | tail -1
for SHAPE in 4096x4096x4096 8192x4096x14336 8192x28672x4096; do
echo "== $SHAPE"
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
    -d /tmp/pmcm_${TAG}_$SHAPE -o pmc -- python $ROOT/scripts/bench_gemm_big.py $SHAPE > /tmp/pmcm_$TAG.log 2>&1 )
tail -3 /tmp/pmcm_$TAG.log
f=$(find /tmp/pmcm_${TAG}_$SHAPE -name "*counter_collection*.csv" | head -1)
echo "counter file: $f"
[ -n "$f" ] && python - "$f" > gpurun_out/pmc_${TAG}_mfma_$SHAPE.json <<'PY'
import csv, json, sys
acc = {}
with open(sys.argv[1], newline="") as fh:
    for row in csv.DictReader(fh):
        k = row["Kernel_Name"]
        if "w4a8_gemm" not in k:
            continue
        short = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        d = acc.setdefault(short, {})
        c = d.setdefault(row["Counter_Name"], [0, 0.0])
        c[0] += 1
        c[1] += float(row["Counter_Value"])
out = {}
for k, d in acc.items():
    m = {c: s / n for c, (n, s) in d.items()}
    m["launches"] = max(n for n, _ in d.values())
    if m.get("SQ_BUSY_CU_CYCLES"):
        # MFMA pipe busy cycles summed over the SIMDs of a CU / (4 SIMDs x CU busy cycles)  [gfx94x MfmaUtil formula]
        m["mfma_util_percent"] = round(100.0 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * m["SQ_BUSY_CU_CYCLES"]), 1)
    out[k] = m
print(json.dumps(out, indent=1))
PY
cat gpurun_out/pmc_${TAG}_mfma_$SHAPE.json | grep -v GRBM | head -20
done
