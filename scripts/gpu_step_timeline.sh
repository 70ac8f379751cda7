#!/bin/bash
# round 5: the decode step as the GPU ran it: kernel durations and the gaps between them inside the replayed hipGraph
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o trace -- python $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-prefill --no-kernel-bench > /tmp/tl.log 2>&1 )
tail -2 /tmp/tl.log | cut -c1-300
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python scripts/step_timeline.py $f 4 | tee gpurun_out/round5_step_timeline.txt
