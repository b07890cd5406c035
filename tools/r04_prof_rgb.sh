#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r04_rgb}; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/profrgb
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profrgb -- python "$ROOT/bench.py" --workload images --rgb --inflight 1 --steps 40 --warmup 5 --no-cpu-baseline --no-stages --no-side-workloads --no-split-line > /tmp/profrgb.log 2>&1
t=$(find /tmp/profrgb -name "*kernel_trace.csv" | head -1)
python "$ROOT/tools/trace_by_grid.py" "$t" > "$OUT/rgb_inflight1_by_grid.md"
grep "enet" "$OUT/rgb_inflight1_by_grid.md" | cut -c1-150
