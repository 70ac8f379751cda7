#!/bin/bash
# Stage the reference's OWN Python (the engine, scheduler, block manager, model code and its benchmark driver) under the
# git-ignored oracle/_ref/ so that it travels to the GPU box with gpurun (the box has no /root/reference):
#   oracle/_ref/qserve/              <- /root/reference/qserve/            (unchanged, byte for byte)
#   oracle/_ref/qserve_benchmark.py  <- /root/reference/qserve_benchmark.py
# Test infrastructure only: nothing under qserve_amd/, qserve_backend*/ imports it (tests/test_abi.py), nothing of it is
# committed (.gitignore: oracle/_ref/).  tests/test_reference_engine_gpu.py and bench.py's reference_engine leg use it when
# present and skip otherwise.
set -eu
REF=${QSERVE_REFERENCE:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
DST=$ROOT/oracle/_ref
[ -d "$REF/qserve" ] || { echo "no reference tree at $REF: nothing staged"; exit 0; }
mkdir -p "$DST"
rm -rf "$DST/qserve"
cp -r "$REF/qserve" "$DST/qserve"
cp "$REF/qserve_benchmark.py" "$DST/qserve_benchmark.py"
find "$DST" -name "__pycache__" -type d -prune -exec rm -rf {} +
( cd "$REF" && find qserve qserve_benchmark.py -type f -name "*.py" | sort | xargs sha256sum ) > "$DST/SHA256SUMS"
echo "staged $(find "$DST/qserve" -name '*.py' | wc -l) reference files under $DST"
