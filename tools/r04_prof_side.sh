#!/bin/bash
# per-launch tables (one chunk at a time) of the side workloads: detect + masks, images (feature maps in), images from RGB
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r04_side}; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for args in "--workload detect --masks" "--workload images" "--workload images --rgb"; do
  i=$((i+1)); rm -rf /tmp/profs$i
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs$i -- python "$ROOT/bench.py" $args --inflight 1 --steps 60 --warmup 10 --no-cpu-baseline --no-stages --no-side-workloads --no-split-line > /tmp/profs$i.log 2>&1
  t=$(find /tmp/profs$i -name "*kernel_trace.csv" | head -1)
  name=$(echo $args | tr -d '-' | tr ' ' '_')
  [ -n "$t" ] && python "$ROOT/tools/trace_by_grid.py" "$t" > "$OUT/${name}_inflight1_by_grid.md"
  grep "^{" /tmp/profs$i.log | tail -1 > "$OUT/${name}_inflight1.json"
done
ls -la "$OUT"
