#!/bin/bash
# the timed decode step under several libraries inside one gpurun call: LIBS="name ..." under _ab_old/ plus "new" (this tree)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
for rep in 1 2; do
  for lib in ${LIBS:-libqserve_amd_r3.so libqserve_amd_prev.so} new; do
    if [ $lib = new ]; then unset QS_AMD_LIBRARY; else export QS_AMD_LIBRARY=$ROOT/_ab_old/$lib; fi
    timeout 600 python bench.py --no-cpu-baseline --no-prefill --no-extras --no-kernel-bench 2>/dev/null > /tmp/b.json
    python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("$lib", "rep $rep", d["value"], d["ms_per_step"])
PY
  done
done
unset QS_AMD_LIBRARY
