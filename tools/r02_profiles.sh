#!/bin/bash
# Round-2 judged artefacts: bench lines for every workload + rocprofv3 kernel stats of the default bench command
set -u
TAG=${1:-r02_final}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
bash "$ROOT/tools/r02_n.sh" "$TAG"
cd /tmp && export TMPDIR=/tmp
for wl in backbone_rpn detect; do
  rm -rf /tmp/prof_$wl
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -- python "$ROOT/bench.py" --workload $wl --steps 100 --warmup 10 --no-cpu-baseline > /tmp/prof_$wl.log 2>&1
  tail -1 /tmp/prof_$wl.log > "$OUT/bench_${wl}_under_rocprof.json"
  f=$(find /tmp/prof_$wl -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/bench_${wl}_kernel_stats.csv"
  t=$(find /tmp/prof_$wl -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python "$ROOT/tools/trace_by_grid.py" "$t" > "$OUT/bench_${wl}_by_grid.md"
  [ -n "$t" ] && [ "$wl" = backbone_rpn ] && python "$ROOT/tools/dominant_from_trace.py" "$t" "$OUT/bench_${wl}_under_rocprof.json" > "$OUT/dominant_kernel_from_trace.json"
  rm -rf /tmp/prof1_$wl
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1_$wl -- python "$ROOT/bench.py" --workload $wl --inflight 1 --steps 100 --warmup 10 --no-cpu-baseline --no-stages > /tmp/prof1_$wl.log 2>&1
  t=$(find /tmp/prof1_$wl -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python "$ROOT/tools/trace_by_grid.py" "$t" > "$OUT/bench_${wl}_inflight1_by_grid.md"
done
head -8 "$OUT/bench_backbone_rpn_by_grid.md" | cut -c1-160
