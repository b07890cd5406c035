#!/bin/bash
# per-launch table of one chunk at a time (no overlap): bash tools/prof_inflight1.sh <tag> [workload]
TAG=${1:-p1}; WL=${2:-backbone_rpn}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof1
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python "$ROOT/bench.py" --workload $WL --inflight 1 --steps 100 --warmup 10 --no-cpu-baseline --no-stages > /tmp/prof1.log 2>&1
tail -1 /tmp/prof1.log | cut -c1-300
t=$(find /tmp/prof1 -name "*kernel_trace.csv" | head -1)
python "$ROOT/tools/trace_by_grid.py" "$t" > "$OUT/${WL}_inflight1_by_grid.md"
head -24 "$OUT/${WL}_inflight1_by_grid.md" | cut -c1-150
