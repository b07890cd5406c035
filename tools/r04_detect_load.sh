#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r04_detect_load}; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/profd
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profd -- python "$ROOT/bench.py" --workload detect --steps 100 --warmup 10 --no-cpu-baseline --no-stages --no-side-workloads --no-split-line --no-live-pmc > /tmp/profd.log 2>&1
t=$(find /tmp/profd -name "*kernel_trace.csv" | head -1)
python "$ROOT/tools/trace_by_grid.py" "$t" > "$OUT/detect_by_grid.md"
python "$ROOT/tools/trace_gaps.py" "$t" > "$OUT/detect_gaps.txt"
head -34 "$OUT/detect_by_grid.md" | cut -c1-150
